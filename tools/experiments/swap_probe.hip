// Probe: semantics of v_permlane32_swap on gfx950 (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(int *o) {
    const int l = threadIdx.x;
    v2i r = __builtin_amdgcn_permlane32_swap(100 + l, 200 + l, false, false);
    o[2 * l] = r.x; o[2 * l + 1] = r.y;
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: x=%d y=%d\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
