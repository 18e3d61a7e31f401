#define PAIR_L 22
#include "em_pair_launch.inc"
