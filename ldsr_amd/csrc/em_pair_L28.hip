#define PAIR_L 28
#include "em_pair_launch.inc"
