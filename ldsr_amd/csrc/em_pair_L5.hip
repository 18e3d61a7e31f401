#define PAIR_L 5
#include "em_pair_launch.inc"
