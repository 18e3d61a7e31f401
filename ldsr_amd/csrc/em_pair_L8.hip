#define PAIR_L 8
#include "em_pair_launch.inc"
