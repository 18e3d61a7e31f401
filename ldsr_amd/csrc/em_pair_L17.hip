#define PAIR_L 17
#include "em_pair_launch.inc"
