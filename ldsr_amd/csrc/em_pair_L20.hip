#define PAIR_L 20
#include "em_pair_launch.inc"
