#define PAIR_L 10
#include "em_pair_launch.inc"
