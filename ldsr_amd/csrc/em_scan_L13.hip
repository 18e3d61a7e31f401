#define SCAN_L 13
#include "em_scan_launch.inc"
