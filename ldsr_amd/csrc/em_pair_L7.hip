#define PAIR_L 7
#include "em_pair_launch.inc"
