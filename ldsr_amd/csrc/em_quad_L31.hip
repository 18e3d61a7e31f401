#define PAIR_L 31
#define PAIR_LPC 16
#include "em_pair_launch.inc"
