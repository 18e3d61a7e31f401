#define SCAN_L 14
#include "em_scan_launch.inc"
