#define SCAN_L 28
#define SCAN_W 4
#include "em_scan_launch.inc"
