#define PAIR_L 30
#define PAIR_LPC 16
#include "em_pair_launch.inc"
