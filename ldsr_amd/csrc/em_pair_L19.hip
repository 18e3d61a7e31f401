#define PAIR_L 19
#include "em_pair_launch.inc"
