// GPU box: which SIMD does wave w of an 8-wave workgroup land on?  (HW_REG_HW_ID: SIMD_ID bits 5:4, WAVE_ID 3:0, CU_ID 11:8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned *out) {
    extern __shared__ double smem[];
    unsigned id = __builtin_amdgcn_s_getreg(63492);     // hwreg(HW_REG_HW_ID, 0, 32)
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    // keep the workgroup resident for a while so that a full grid co-resides like the EM kernel's
    double x = threadIdx.x; for (int i = 0; i < 20000; i++) x = fma(x, 0.999, 1e-3); if (x == 1.2345) out[0] = 0;
}
int main() {
    const int nb = 256;
    unsigned *d; hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    probe<<<nb, 512, 100 * 1024>>>(d);
    unsigned h[nb * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < nb; b++) for (int w = 0; w < 8; w++) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; w++) printf("wave %d of the workgroup: SIMD 0..3 counts %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; b++) { printf("block %d:", b); for (int w = 0; w < 8; w++) printf(" (simd %u slot %u cu %u)", (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
    return 0;
}
