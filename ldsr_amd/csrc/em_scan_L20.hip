#define SCAN_L 20
#include "em_scan_launch.inc"
