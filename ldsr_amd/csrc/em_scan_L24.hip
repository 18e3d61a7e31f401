#define SCAN_L 24
#include "em_scan_launch.inc"
