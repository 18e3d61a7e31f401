#define PAIR_L 16
#define PAIR_LPC 64
#include "em_pair_launch.inc"
