#define SCAN_L 32
#include "em_scan_launch.inc"
