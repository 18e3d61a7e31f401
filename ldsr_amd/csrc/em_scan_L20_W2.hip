#define SCAN_L 20
#define SCAN_W 2
#include "em_scan_launch.inc"
