// GPU box: hipcc --offload-arch=gfx950 -O3 tools/experiments/lone_wave_probe.hip -o /tmp/lone_wave_probe && /tmp/lone_wave_probe
// How fast does ONE wave issue dependent fp64 work?  N interleaved chains of v_fma_f64 (N = 1, 2, 3, 4, 8), a chain of
// DPP moves + fma, a chain through v_readlane, v_rcp_f64 / v_ldexp_f64 chains: shader cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N>
__global__ void fma_chains(double *out, long long *cyc, double a, double b) {
    double x[N];
    for (int i = 0; i < N; i++) x[i] = threadIdx.x * 1e-3 + i;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = fma(x[i], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < N; i++) s += x[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void dpp_chain(double *out, long long *cyc, double a) {
    double x = threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xF, 0xF, true);   // row_shr:1
            int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xF, 0xF, true);
            x = fma(__hiloint2double(hi, lo), a, x);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void readlane_chain(double *out, long long *cyc, double a) {
    double x = threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            int lo = __builtin_amdgcn_readlane(__double2loint(x), 5);
            int hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
            x = fma(__hiloint2double(hi, lo), a, x);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void rcp_chain(double *out, long long *cyc) {
    double x = 1.5 + threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = __builtin_amdgcn_rcp(x);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void ldexp_chain(double *out, long long *cyc, int e) {
    double x = 1.5 + threadIdx.x * 1e-3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = __builtin_amdgcn_ldexp(x, e);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void lds_chain(double *out, long long *cyc) {
    __shared__ double buf[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += 64) buf[i] = 0.0;
    __syncthreads();
    double x = 0.0;
    int idx = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int k = 0; k < 32; k++) { x += buf[idx + (int)x]; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    long long h;
    auto rep = [&](const char *name, int insts) {
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %8lld cycles  %6.2f per instruction (%d instructions)\n", name, h, (double)h / insts, insts);
    };
    for (int w = 0; w < 2; w++) {   // second round: warm
        fma_chains<1><<<1, 64>>>(out, cyc, 0.999, 1e-3); if (w) rep("1 chain of v_fma_f64", 2048);
        fma_chains<2><<<1, 64>>>(out, cyc, 0.999, 1e-3); if (w) rep("2 interleaved chains", 4096);
        fma_chains<3><<<1, 64>>>(out, cyc, 0.999, 1e-3); if (w) rep("3 interleaved chains", 6144);
        fma_chains<4><<<1, 64>>>(out, cyc, 0.999, 1e-3); if (w) rep("4 interleaved chains", 8192);
        fma_chains<8><<<1, 64>>>(out, cyc, 0.999, 1e-3); if (w) rep("8 interleaved chains", 16384);
        dpp_chain<<<1, 64>>>(out, cyc, 0.5); if (w) rep("chain: 2 DPP moves + 1 fma (3 insts / link)", 2048 * 3);
        readlane_chain<<<1, 64>>>(out, cyc, 0.5); if (w) rep("chain: 2 v_readlane + 1 fma (3 insts / link)", 2048 * 3);
        rcp_chain<<<1, 64>>>(out, cyc); if (w) rep("chain of v_rcp_f64", 2048);
        ldexp_chain<<<1, 64>>>(out, cyc, 1); if (w) rep("chain of v_ldexp_f64", 2048);
        lds_chain<<<1, 64>>>(out, cyc); if (w) rep("chain: ds_read_b64 -> add (dependent address)", 2048);
    }
    // two waves on one SIMD? (blocks of 512 threads = 8 waves = 2 per SIMD)
    fma_chains<1><<<1, 512>>>(out, cyc, 0.999, 1e-3); rep("1 chain, 8 waves in the workgroup (2 / SIMD)", 2048);
    fma_chains<2><<<1, 512>>>(out, cyc, 0.999, 1e-3); rep("2 chains, 8 waves in the workgroup", 4096);
    return 0;
}
