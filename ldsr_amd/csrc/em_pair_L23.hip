#define PAIR_L 23
#include "em_pair_launch.inc"
