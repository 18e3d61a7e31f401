#define PAIR_L 13
#include "em_pair_launch.inc"
