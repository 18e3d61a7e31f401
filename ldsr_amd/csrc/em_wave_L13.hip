#define PAIR_L 13
#define PAIR_LPC 64
#include "em_pair_launch.inc"
