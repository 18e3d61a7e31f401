#define PAIR_L 9
#include "em_pair_launch.inc"
