#define PAIR_L 24
#include "em_pair_launch.inc"
