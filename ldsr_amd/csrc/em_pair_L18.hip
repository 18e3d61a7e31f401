#define PAIR_L 18
#include "em_pair_launch.inc"
