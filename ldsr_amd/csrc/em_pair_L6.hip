#define PAIR_L 6
#include "em_pair_launch.inc"
