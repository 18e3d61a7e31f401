#define PAIR_L 26
#include "em_pair_launch.inc"
