#define SCAN_L 10
#include "em_scan_launch.inc"
