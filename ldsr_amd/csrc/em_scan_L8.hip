#define SCAN_L 8
#include "em_scan_launch.inc"
