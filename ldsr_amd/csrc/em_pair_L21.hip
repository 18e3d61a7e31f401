#define PAIR_L 21
#include "em_pair_launch.inc"
