#define SCAN_L 16
#include "em_scan_launch.inc"
