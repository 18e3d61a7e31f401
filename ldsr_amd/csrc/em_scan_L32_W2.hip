#define SCAN_L 32
#define SCAN_W 2
#include "em_scan_launch.inc"
