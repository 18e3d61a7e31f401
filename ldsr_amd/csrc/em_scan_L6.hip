#define SCAN_L 6
#include "em_scan_launch.inc"
