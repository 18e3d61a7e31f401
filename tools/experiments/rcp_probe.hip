// Probe: accuracy of v_rcp_f64 + k Newton steps on gfx950 (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* r3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = fma(-v, r, 1.0); r = fma(r, e, r); r1[i] = r;
    e = fma(-v, r, 1.0); r = fma(r, e, r); r2[i] = r;
    // one third-order step from the seed: r (1 + e + e^2)
    r = r0[i]; e = fma(-v, r, 1.0); r3[i] = fma(r, fma(e, e, e), r);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n), d(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (s >> 11) * (1.0 / 9007199254740992.0), (int)(s % 41) - 20); }
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0; long ne1 = 0, ne2 = 0, ne3 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        double ex = (double)t;
        m0 = fmax(m0, fabs((double)((a[i] - t) / t)));
        m1 = fmax(m1, fabs((double)((b[i] - t) / t)));
        m2 = fmax(m2, fabs((double)((c[i] - t) / t)));
        m3 = fmax(m3, fabs((double)((d[i] - t) / t)));
        ne1 += b[i] != ex; ne2 += c[i] != ex; ne3 += d[i] != ex;
    }
    printf("max rel err: rcp %.3g, +1NR %.3g (%.2f%% != correctly rounded), +2NR %.3g (%.2f%%)\n", m0, m1, 100.0 * ne1 / n, m2, 100.0 * ne2 / n);
    printf("third-order step (3 FMAs): max rel err %.3g (%.2f%% != correctly rounded)\n", m3, 100.0 * ne3 / n);
    return 0;
}
