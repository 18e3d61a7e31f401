#define SCAN_L 12
#include "em_scan_launch.inc"
