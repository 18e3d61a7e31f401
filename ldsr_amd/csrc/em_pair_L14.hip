#define PAIR_L 14
#include "em_pair_launch.inc"
