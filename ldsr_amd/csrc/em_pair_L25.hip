#define PAIR_L 25
#include "em_pair_launch.inc"
