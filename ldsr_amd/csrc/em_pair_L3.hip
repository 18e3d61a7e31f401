#define PAIR_L 3
#include "em_pair_launch.inc"
