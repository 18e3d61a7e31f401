#define SCAN_L 2
#include "em_scan_launch.inc"
