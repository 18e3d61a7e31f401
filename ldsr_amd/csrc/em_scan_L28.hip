#define SCAN_L 28
#include "em_scan_launch.inc"
