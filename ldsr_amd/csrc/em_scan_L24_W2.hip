#define SCAN_L 24
#define SCAN_W 2
#include "em_scan_launch.inc"
