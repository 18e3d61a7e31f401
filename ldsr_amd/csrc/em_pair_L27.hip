#define PAIR_L 27
#include "em_pair_launch.inc"
