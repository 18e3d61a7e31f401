#define SCAN_L 15
#include "em_scan_launch.inc"
