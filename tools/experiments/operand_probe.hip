// Probe: does the issue rate of fp64 VALU instructions on gfx950 depend on where the operands come
// from (VGPR x3, VGPR x2 + SGPR, accumulate form)?  8 independent chains, 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/operand_probe.hip -o tools/operand_probe && tools/operand_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NCH 8
#define REP 64

template <int KIND>
__global__ __launch_bounds__(512) void k(double *out, const double *in, int iters, double a, double b) {
    double x[NCH], y[NCH], z[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        x[c] = in[threadIdx.x + 64 * c]; y[c] = in[threadIdx.x + 64 * c + 1024]; z[c] = in[threadIdx.x + 64 * c + 2048];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (KIND == 0) x[c] = fma(x[c], a, b);                 // VGPR, SGPR, SGPR
                if (KIND == 1) x[c] = fma(x[c], y[c], b);              // VGPR, VGPR, SGPR
                if (KIND == 2) x[c] = fma(x[c], y[c], z[c]);           // three VGPRs, dst = src0
                if (KIND == 3) x[c] = fma(y[c], z[c], x[c]);           // v_fmac: dst = src2
                if (KIND == 4) x[c] = fma(y[c], z[(c + 1) % NCH], x[(c + 3) % NCH]);  // three VGPRs, dst different
                if (KIND == 5) x[c] = x[c] * y[c];                     // v_mul VGPR, VGPR
                if (KIND == 6) x[c] = x[c] + y[c];                     // v_add VGPR, VGPR
                if (KIND == 7) x[c] = fma(-x[c], y[c], z[c]);          // source modifier -> VOP3
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c];
    if (s == 12345.678) out[0] = s;
}

template <int KIND>
static void run(const char *name) {
    double *out, *in;
    hipMalloc(&out, 8); hipMalloc(&in, 8 * 4096); hipMemset(in, 0, 8 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2}) {
        const int threads = 64 * 4 * wps, blocks = 256, iters = 200;
        k<KIND><<<blocks, threads>>>(out, in, 2, 1.0000001, 1e-9);
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(out, in, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)iters * REP * NCH * wps;
        printf("%-40s %d waves/SIMD: %.2f ns per wave-instruction per SIMD\n", name, wps, ms * 1e6 / winst);
    }
    hipFree(out); hipFree(in);
}

int main() {
    run<0>("fma  v, s, s");
    run<1>("fma  v, v, s");
    run<2>("fma  v, v, v (dst = src0)");
    run<3>("fmac v, v, acc");
    run<4>("fma  v, v, v (dst elsewhere)");
    run<5>("mul  v, v");
    run<6>("add  v, v");
    run<7>("fma -v, v, v");
    return 0;
}
