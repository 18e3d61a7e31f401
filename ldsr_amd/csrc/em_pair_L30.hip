#define PAIR_L 30
#include "em_pair_launch.inc"
