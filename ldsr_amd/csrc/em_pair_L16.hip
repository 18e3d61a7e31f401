#define PAIR_L 16
#include "em_pair_launch.inc"
