#define SCAN_L 20
#define SCAN_W 4
#include "em_scan_launch.inc"
