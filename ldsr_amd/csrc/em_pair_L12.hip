#define PAIR_L 12
#include "em_pair_launch.inc"
