#define PAIR_L 15
#include "em_pair_launch.inc"
