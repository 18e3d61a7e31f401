#define PAIR_L 29
#include "em_pair_launch.inc"
