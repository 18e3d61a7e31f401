// kernels_scan.hip -- chooses the chunk length L (time steps per lane) of the wave-per-cell
// scan kernel and dispatches to the per-L translation units (em_scan_L*.hip).
#include "em_scan_impl.h"
#include <cstdio>

#include "ldsr_kernels.h"

// smallest compiled chunk length with T <= 64*L; every choice also satisfies L*(L-1) <= T
static int scan_L_for(int T) {
    static const int Ls[] = {2, 3, 4, 6, 8, 10, 12, 13, 14, 15, 16, 20, 24, 28, 32};
    for (int L : Ls)
        if (T <= 64 * L) return (T >= L * (L - 1)) ? L : 0;
    return 0;
}

static bool lds_image_fits(int L, int PP, int QQ) {
    return (size_t)64 * L * (1 + PP + QQ) * sizeof(double) <= 160 * 1024;
}

bool em_scan_supported(int T, int PP, int QQ) {
    const int L = scan_L_for(T);
    if (!L || PP > 8 || QQ > 8) return false;   // instantiated for padded widths up to 8
    return lds_image_fits(L, PP, QQ) || L >= 20; // L >= 20 also exists in a global-image variant
}

// Long series with wide inputs: the chunk-transposed LDS image would exceed 160 KiB; the kernel
// variant GIMG reads the prepared time-major arrays straight from global memory (L2 resident).
bool em_scan_global_image(int T, int PP, int QQ) {
    const int L = scan_L_for(T);
    return L >= 20 && PP <= 8 && QQ <= 8 && !lds_image_fits(L, PP, QQ);
}

int em_scan_cells_per_block(int T, int PP, int QQ) {
    if (em_scan_global_image(T, PP, QQ)) return 4;     // no LDS image: two workgroups per CU by VGPRs
    return scan_wpb(scan_L_for(T), PP, QQ);
}

bool em_scan_queue_only(int T, int PP, int QQ) { return em_scan_global_image(T, PP, QQ); }

void em_scan_kernel_name(int T, int PP, int QQ, bool queue, char *buf, size_t len) {
    const bool gimg = em_scan_global_image(T, PP, QQ);
    snprintf(buf, len, "em_scan_kernel<%d, %d, %d, %s, %s>", PP, QQ, scan_L_for(T),
             (queue || gimg) ? "true" : "false", gimg ? "true" : "false");
}

hipError_t launch_em_scan(const EmParams &prm, int PP, int QQ, int n_blocks, bool queue,
                          hipStream_t stream) {
    const int wpb = em_scan_cells_per_block(prm.T, PP, QQ);
    if (em_scan_global_image(prm.T, PP, QQ)) queue = true;   // GIMG exists with the queue schedule only
    switch (scan_L_for(prm.T)) {
        case 2: return launch_em_scan_L<2>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 3: return launch_em_scan_L<3>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 4: return launch_em_scan_L<4>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 6: return launch_em_scan_L<6>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 8: return launch_em_scan_L<8>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 10: return launch_em_scan_L<10>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 12: return launch_em_scan_L<12>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 13: return launch_em_scan_L<13>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 14: return launch_em_scan_L<14>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 15: return launch_em_scan_L<15>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 16: return launch_em_scan_L<16>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 20: return launch_em_scan_L<20>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 24: return launch_em_scan_L<24>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 28: return launch_em_scan_L<28>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        case 32: return launch_em_scan_L<32>(prm, PP, QQ, n_blocks, wpb, queue, stream);
        default: return hipErrorInvalidValue;
    }
}
