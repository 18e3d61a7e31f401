#define PAIR_L 4
#include "em_pair_launch.inc"
