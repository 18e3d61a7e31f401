#define PAIR_L 31
#include "em_pair_launch.inc"
