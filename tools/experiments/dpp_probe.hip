// Probe: semantics of the DPP controls used by the scan kernel on gfx950 (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int RM>
__device__ __forceinline__ double dppd(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, RM, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, RM, 0xF, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* o) {
    double x = (double)threadIdx.x;
    o[threadIdx.x * 8 + 0] = dppd<0x111, 0xF>(-1.0, x);  // row_shr:1
    o[threadIdx.x * 8 + 1] = dppd<0x118, 0xF>(-1.0, x);  // row_shr:8
    o[threadIdx.x * 8 + 2] = dppd<0x142, 0xA>(-1.0, x);  // row_bcast15 rows 1,3
    o[threadIdx.x * 8 + 3] = dppd<0x143, 0xC>(-1.0, x);  // row_bcast31 rows 2,3
    o[threadIdx.x * 8 + 4] = dppd<0x138, 0xF>(-1.0, x);  // wave_shr:1
    o[threadIdx.x * 8 + 5] = dppd<0x130, 0xF>(-1.0, x);  // wave_shl:1
    o[threadIdx.x * 8 + 6] = dppd<0x101, 0xF>(-1.0, x);  // row_shl:1
    o[threadIdx.x * 8 + 7] = dppd<0x104, 0xF>(-1.0, x);  // row_shl:4
}
int main() {
    double* d; (void)hipMalloc(&d, 64 * 8 * 8);
    k<<<1, 64>>>(d);
    double h[512]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[] = {"row_shr1", "row_shr8", "bcast15/A", "bcast31/C", "wave_shr1", "wave_shl1", "row_shl1", "row_shl4"};
    for (int c = 0; c < 8; c++) { printf("%-10s", nm[c]); for (int l = 0; l < 64; l++) printf(" %g", h[l * 8 + c]); printf("\n"); }
    return 0;
}
