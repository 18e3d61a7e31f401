// Probe: dependent-issue latency of the instruction kinds on the EM kernels' critical paths (gfx950):
// NCH independent dependency chains per lane (1..4) at 1 and 2 waves per SIMD.  With one chain the
// time per instruction IS the dependent-issue latency; the knee tells how much ILP a wave needs.
//   hipcc -O3 --offload-arch=gfx950 tools/latency_probe.hip -o tools/latency_probe && tools/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64

template <int KIND, int NCH>
__global__ __launch_bounds__(512) void k(double *out, int iters, double a, double b) {
    __shared__ double sm[1024];
    double x[NCH];
    int xi[NCH];
    sm[threadIdx.x] = threadIdx.x & 63;
    sm[512 + threadIdx.x] = threadIdx.x & 63;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; c++) { x[c] = a + threadIdx.x * 1e-9 + c; xi[c] = (threadIdx.x + c) & 63; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (KIND == 0) x[c] = fma(x[c], a, b);                       // dependent v_fma_f64
                if (KIND == 1) {                                             // fma -> DPP mov (2 words) -> fma
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[c]), 0x111, 0xF, 0xF, true);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x[c]), 0x111, 0xF, 0xF, true);
                    x[c] = fma(__hiloint2double(hi, lo), a, b);
                }
                if (KIND == 2) xi[c] = __builtin_amdgcn_ds_bpermute(xi[c] << 2, xi[c] + 1) & 63;   // dependent ds_bpermute
                if (KIND == 3) xi[c] = (int)sm[(xi[c] + threadIdx.x) & 1023] ;          // dependent ds_read_b64 (+cvt)
                if (KIND == 4) x[c] = __builtin_amdgcn_rcp(x[c]);            // dependent v_rcp_f64
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c] + xi[c];
    if (s == 12345.678) out[0] = s;
}

template <int KIND, int NCH>
static void run1(const char *name, int inst) {
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2}) {
        const int threads = 64 * 4 * wps, blocks = 256, iters = 100;
        k<KIND, NCH><<<blocks, threads>>>(out, 2, 1.0000001, 1e-9);
        hipEventRecord(e0);
        k<KIND, NCH><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double steps = (double)iters * REP;          // chain steps per wave
        printf("%-34s chains %d, %d waves/SIMD: %.2f ns per chain step per wave (%d instr each), %.2f ns per wave-instr per SIMD\n",
               name, NCH, wps, ms * 1e6 / steps, inst, ms * 1e6 / (steps * NCH * inst * wps));
    }
    hipFree(out);
}
template <int KIND>
static void run(const char *name, int inst) {
    run1<KIND, 1>(name, inst); run1<KIND, 2>(name, inst); run1<KIND, 3>(name, inst); run1<KIND, 4>(name, inst);
}

int main() {
    run<0>("v_fma_f64 dependent", 1);
    run<1>("fma -> dpp movs -> fma", 3);
    run<2>("ds_bpermute dependent", 1);
    run<3>("ds_read_b64 dependent", 1);
    run<4>("v_rcp_f64 dependent", 1);
    return 0;
}
