#define PAIR_L 11
#include "em_pair_launch.inc"
