#define SCAN_L 4
#include "em_scan_launch.inc"
