#define SCAN_L 3
#include "em_scan_launch.inc"
