#define PAIR_L 21
#define PAIR_LPC 16
#include "em_pair_launch.inc"
