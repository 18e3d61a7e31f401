#define PAIR_L 32
#include "em_pair_launch.inc"
