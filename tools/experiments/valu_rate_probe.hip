// Probe: sustained issue rate of the VALU instruction kinds the EM kernels are made of, on gfx950,
// at 1 / 2 / 4 waves per SIMD (diagnostic, not part of the library).  Each kernel runs NCH
// independent dependency chains per lane so that latency is covered inside one wave as well.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe && tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NCH 8
#define REP 64

template <int KIND>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double a, double b) {
    double x[NCH];
    int xi[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { x[c] = a + threadIdx.x * 1e-9 + c; xi[c] = threadIdx.x + c; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (KIND == 0) x[c] = fma(x[c], a, b);                       // v_fma_f64
                if (KIND == 1) x[c] = x[c] * a;                              // v_mul_f64
                if (KIND == 2) x[c] = x[c] + b;                              // v_add_f64
                if (KIND == 3) xi[c] = __builtin_amdgcn_update_dpp(xi[c], xi[c], 0x111, 0xF, 0xF, false) + 0;  // v_mov_b32 dpp row_shr:1
                if (KIND == 4) xi[c] = (xi[c] ^ (int)it) + c;                // 32-bit integer pair (xor + add)
                if (KIND == 5) x[c] = __builtin_amdgcn_rcp(x[c]);            // v_rcp_f64
                if (KIND == 6) { x[c] = fma(x[c], a, b); xi[c] = __builtin_amdgcn_update_dpp(xi[c], xi[c], 0x111, 0xF, 0xF, false); }  // fma + dpp mov interleaved
                if (KIND == 7) { x[c] = fma(x[c], a, b); xi[c] = xi[c] > r ? xi[c] : c; }   // fma + v_cndmask-ish
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c] + xi[c];
    if (s == 12345.678) out[0] = s;
}

template <int KIND>
static void run(const char *name, int inst_per_slot) {
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;      // 4 SIMDs per CU
        const int blocks = 256;
        const int iters = 200;
        k<KIND><<<blocks, threads>>>(out, 2, 1.0000001, 1e-9);
        hipEventRecord(e0);
        k<KIND><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)iters * REP * NCH * inst_per_slot * wps;   // wave-instructions per SIMD
        printf("%-28s %d waves/SIMD: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
               name, wps, ms, ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f64", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_add_f64", 1);
    run<3>("v_mov_b32 dpp", 1);
    run<4>("int32 xor+add", 2);
    run<5>("v_rcp_f64", 1);
    run<6>("fma_f64 + dpp mov", 2);
    run<7>("fma_f64 + cmp/cndmask", 3);
    return 0;
}
