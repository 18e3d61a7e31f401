// kernels_scan.hip -- launch plan of the parallel-in-time scan kernel (chunk length L, waves
// per cell W, LDS or global series image) and dispatch to the per-(L, W) translation units
// (em_scan_L*.hip).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "em_scan_impl.h"
#include "em_pair_impl.h"
#include "ldsr_kernels.h"

static const size_t kLdsBytes = 160 * 1024;

// waves per block so that a CU holds ~8 waves given the LDS image of one series
static int scan_wpb(int L, int PP, int QQ) {
    if (scan_wide(PP, QQ)) return 4;      // one wave per SIMD (512-register kernels)
    const size_t lds = (size_t)scan_image_doubles(L, 1, PP, QQ) * sizeof(double);
    const int blocks_per_cu = (int)(kLdsBytes / lds);
    const int want = 8;
    int wpb = (want + blocks_per_cu - 1) / (blocks_per_cu > 0 ? blocks_per_cu : 1);
    if (wpb < 2) wpb = 2;
    if (L <= 16 && wpb < 4 && lds > 20 * 1024) wpb = 4;
    const int cap = 8;
    return wpb > cap ? cap : wpb;
}

// W = 1: smallest compiled chunk length with T <= 64 L (every choice also has L (L-1) <= T).
// T > 2048: W = 2 (T <= 4096) or 4 (T <= 8192) waves per cell with half-stored chunks.
ScanPlan scan_plan(int T, int PP, int QQ) {
    ScanPlan p;
    if (PP > 8 || QQ > 8 || T < 2) return p;   // instantiated for padded widths up to 8
    static const int Ls[] = {2, 3, 4, 6, 8, 10, 12, 13, 14, 15, 16, 20, 24, 28, 32};
    static const int LsW[] = {20, 24, 28, 32};
    if (T <= 2048) {
        p.W = 1;
        for (int L : Ls)
            if (T <= 64 * L) { p.L = (T >= L * (L - 1)) ? L : 0; break; }
    } else if (T <= 8192) {
        p.W = T <= 4096 ? 2 : 4;
        for (int L : LsW)
            if (T <= 64 * p.W * L) { p.L = L; break; }
    }
    if (!p.L) return p;
    // the image is read from global memory (raw buffer loads, L2 / L1 resident) when it does not
    // fit a CU's LDS (L >= 20 only: the short-chunk images always fit) and for every multi-wave
    // cell: one cell per workgroup would let an LDS image cap a CU at one or two cells, while the
    // global image leaves 4 (W = 2) or 2 (W = 4) cells per CU -- measured 1.5x faster at W = 2
    p.gimg = scan_uses_gimg(p.L, p.W, PP, QQ);
    if (!p.gimg && !scan_image_fits_lds(p.L, p.W, PP, QQ)) return p;
    p.cpb = p.W > 1 ? 1 : (p.gimg ? 4 : scan_wpb(p.L, PP, QQ));   // GIMG: two workgroups per CU by VGPRs
    p.ok = true;
    return p;
}

bool em_scan_supported(int T, int PP, int QQ) { return scan_plan(T, PP, QQ).ok; }
bool em_scan_global_image(int T, int PP, int QQ) { return scan_plan(T, PP, QQ).gimg; }
int em_scan_cells_per_block(int T, int PP, int QQ) { return scan_plan(T, PP, QQ).cpb; }
// shapes compiled with the work-queue schedule only: global-image and multi-wave kernels
bool em_scan_queue_only(int T, int PP, int QQ) {
    const ScanPlan p = scan_plan(T, PP, QQ);
    return p.gimg || p.W > 1;
}
void em_scan_layout(int T, int PP, int QQ, int *L, int *NL, long *img_doubles) {
    const ScanPlan p = scan_plan(T, PP, QQ);
    *L = p.L;
    *NL = 64 * p.W;
    *img_doubles = p.ok ? scan_image_doubles(p.L, p.W, PP, QQ) : 0;
}

void em_scan_kernel_name(int T, int PP, int QQ, bool queue, bool fit, char *buf, size_t len) {
    const ScanPlan p = scan_plan(T, PP, QQ);
    queue = !fit && (queue || p.gimg || p.W > 1);
    snprintf(buf, len, "em_scan_kernel<%d, %d, %d, %d, %s, %s, %s>", PP, QQ, p.L, p.W,
             queue ? "true" : "false", p.gimg ? "true" : "false", fit ? "true" : "false");
}

hipError_t launch_em_scan(const EmParams &prm, int PP, int QQ, int n_blocks, bool queue, bool fit,
                          hipStream_t stream) {
    const ScanPlan p = scan_plan(prm.T, PP, QQ);
    if (!p.ok) return hipErrorInvalidValue;
    queue = !fit && (queue || p.gimg || p.W > 1);
#define CASE_LW(Lv, Wv) \
    case Lv * 8 + Wv: return launch_em_scan_LW<Lv, Wv>(prm, PP, QQ, n_blocks, p.cpb, queue, p.gimg, fit, stream);
    switch (p.L * 8 + p.W) {
        CASE_LW(2, 1) CASE_LW(3, 1) CASE_LW(4, 1) CASE_LW(6, 1) CASE_LW(8, 1) CASE_LW(10, 1)
        CASE_LW(12, 1) CASE_LW(13, 1) CASE_LW(14, 1) CASE_LW(15, 1) CASE_LW(16, 1)
        CASE_LW(20, 1) CASE_LW(24, 1) CASE_LW(28, 1) CASE_LW(32, 1)
        CASE_LW(20, 2) CASE_LW(24, 2) CASE_LW(28, 2) CASE_LW(32, 2)
        CASE_LW(20, 4) CASE_LW(24, 4) CASE_LW(28, 4) CASE_LW(32, 4)
        default: return hipErrorInvalidValue;
    }
#undef CASE_LW
}

// ---- two / four cells per wave (em_pair_impl.h) --------------------------------------------------
// lpc = lanes per cell: 32 (two cells per wave, T <= 1024) or 16 (four, T <= 512).
// Smallest chunk length with L (L-1) <= T <= lpc L.  The kernel pays off only with two waves per
// SIMD: eight waves per workgroup (one workgroup fills a CU's LDS), i.e. the lpc-lane series image
// and eight strips must fit 160 KiB -- at lpc = 32: (1,2): every L; (1,4): L <= 29; (2,4), (4,2):
// L <= 27; (4,4): L <= 25.  Same-box A/B at 8192 cells: with 7 or 6 waves per CU (and the coarser
// workgroup count) it is 20-25 % SLOWER than the one-cell-per-wave kernel, with 8 it is 10-15 % faster.
PairPlan pair_plan(int T, int PP, int QQ, int lpc, bool lead_form) {
    PairPlan p;
    p.lpc = lpc;
    // Wide inputs (padded p or q = 8) exist as the two-cells-per-wave LEAD form only: the tail of a
    // closed-form lead in chunks of <= 16 steps (the generic sweeps of long chunks do not fit the
    // registers of two waves per SIMD at q = 8; the lead itself never touches v_t).
    const bool wide = PP > 4 || QQ > 4;
    if ((lpc != 32 && lpc != 16) || PP > 8 || QQ > 8 || T <= 64) return p;
    if (wide && !(lead_form && ((lpc == 32 && T <= 512) || (lpc == 16 && T <= 256)))) return p;
    if (lead_form && T > lpc * 16) return p;      // (LEAD forms: chunks of <= 16 steps)
    // every chunk length from 3 to 32: the shortest one wastes no lanes (four cells per wave: from 5)
    // (lanes 0 .. rp-1 own L steps, the others L-1: needs 1 <= rp <= nl)
    for (int L = (lpc == 16 ? 5 : 3); L <= 32; L++) {
        if (T > lpc * L) continue;
        const int nl = (T + L - 1) / L, rp = T - nl * (L - 1);
        if (rp >= 1 && rp <= nl) { p.L = L; break; }
    }
    if (!p.L) return p;
    if (!pair_member_fits(p.L, lpc, PP, QQ)) return p;     // (wide: four waves per workgroup may have to do)
    p.wpb = 8;
    p.ok = true;
    return p;
}

// Waves per workgroup of the launch: a CU holds eight waves of these kernels (256 VGPRs each), as
// one workgroup of eight or -- where the LDS has room for two images and two sets of strips (and
// two copies of a lead's u_t) -- as two workgroups of four.  The finer grain fills the last round
// of a launch better: config 4 (10 240 cells = 320 eight-wave workgroups on 256 CUs: two rounds,
// the second on a quarter of the device) 4.56 -> 3.45 ms, config 5 (768 = three full rounds)
// 3.42 -> 3.29 ms; config 2's image allows one workgroup per CU only.  LDSR_PAIR_WPB=8 keeps
// eight (A/B hook).
int em_pair_waves_per_block(int T, int PP, int QQ, int lpc, int lead) {
    const PairPlan p = pair_plan(T, PP, QQ, lpc, lead > 0);
    if (!p.ok) return 0;
    static const int wpb_env = [] { const char *e = getenv("LDSR_PAIR_WPB"); return e ? atoi(e) : 0; }();
    if (wpb_env == 8) return 8;
    const int w = wpb_env == 2 ? 2 : 4;
    const size_t lds = ((size_t)pair_image_doubles(p.L, PP, QQ, lpc) + (size_t)w * pair_strip_doubles(p.L) +
                        (lead > 0 ? (size_t)pair_lead_doubles(lead, lpc, PP) : 0) +
                        (lead == 0 && pair_steady(p.L, lpc, PP, QQ) ? (size_t)pair_tri_doubles(PP, QQ, lpc) : 0)) * sizeof(double);
    // (the runtime keeps some LDS per workgroup for itself: leave 1 KiB per workgroup free)
    if ((8 / w) * (lds + 1024) <= kLdsBytes) return w;
    // (wide LEAD forms whose image, strips and lead leave no room for eight waves: ONE workgroup of four)
    const size_t lds8 = lds + (size_t)(8 - w) * pair_strip_doubles(p.L) * sizeof(double);
    return lds8 <= kLdsBytes ? 8 : w;
}

bool em_pair_supported(int T, int PP, int QQ, int lpc, bool lead_form) { return pair_plan(T, PP, QQ, lpc, lead_form).ok; }
int em_pair_cells_per_block(int T, int PP, int QQ, int lpc, int lead) { return (64 / lpc) * em_pair_waves_per_block(T, PP, QQ, lpc, lead); }
void em_pair_layout(int T, int PP, int QQ, int lpc, int *L, long *img_doubles, bool lead_form) {
    const PairPlan p = pair_plan(T, PP, QQ, lpc, lead_form);
    *L = p.L;
    *img_doubles = p.ok ? pair_image_doubles(p.L, PP, QQ, lpc) : 0;
}
void em_pair_kernel_name(int T, int PP, int QQ, int lpc, bool queue, char *buf, size_t len, bool lead) {
    const PairPlan p = pair_plan(T, PP, QQ, lpc, lead);
    snprintf(buf, len, "em_pair_kernel<%d, %d, %d, %d, %s, %s>", PP, QQ, p.L, lpc, queue ? "true" : "false", lead ? "true" : "false");
}

hipError_t launch_em_pair(const EmParams &prm, int PP, int QQ, int lpc, int n_blocks, bool queue,
                          hipStream_t stream) {
    PairPlan p = pair_plan(prm.T - prm.lead, PP, QQ, lpc, prm.lead > 0);     // (LEAD form: the tail's plan)
    if (!p.ok || !prm.img2 || (prm.lead > 0 && !prm.img3)) return hipErrorInvalidValue;
    p.wpb = em_pair_waves_per_block(prm.T - prm.lead, PP, QQ, lpc, prm.lead);
#define CASE_L(Lv) case Lv: return lpc == 32 ? launch_em_pair_L<Lv, 32>(prm, PP, QQ, n_blocks, p.wpb, queue, stream) \
                                             : launch_em_pair_L<Lv, 16>(prm, PP, QQ, n_blocks, p.wpb, queue, stream);
    switch (p.L) {
        case 3: return launch_em_pair_L<3, 32>(prm, PP, QQ, n_blocks, p.wpb, queue, stream);
        case 4: return launch_em_pair_L<4, 32>(prm, PP, QQ, n_blocks, p.wpb, queue, stream);
        CASE_L(5) CASE_L(6) CASE_L(7) CASE_L(8) CASE_L(9) CASE_L(10) CASE_L(11) CASE_L(12) CASE_L(13) CASE_L(14)
        CASE_L(15) CASE_L(16) CASE_L(17) CASE_L(18) CASE_L(19) CASE_L(20) CASE_L(21) CASE_L(22) CASE_L(23)
        CASE_L(24) CASE_L(25) CASE_L(26) CASE_L(27) CASE_L(28) CASE_L(29) CASE_L(30) CASE_L(31) CASE_L(32)
        default: return hipErrorInvalidValue;
    }
#undef CASE_L
}


// ---- what is compiled ------------------------------------------------------------------------------
// Every instantiation of the scan and pair families this library holds, by the names rocprofv3 prints,
// derived from the SAME predicates the launchers instantiate with (em_scan_launch.inc launch_one,
// em_pair_launch.inc): tests/test_abi_and_host.py checks that this set equals what the launch plans can
// return over the supported domain -- nothing unreachable is compiled, nothing reachable is missing.
void em_kernel_inventory(std::string &out) {
    static const int widths[] = {1, 2, 4, 8};
    static const int Ls1[] = {2, 3, 4, 6, 8, 10, 12, 13, 14, 15, 16, 20, 24, 28, 32};
    static const int LsW[] = {20, 24, 28, 32};
    char buf[160];
    auto scan_names = [&](int L, int W) {
        for (int PP : widths)
            for (int QQ : widths) {
                const bool gimg = scan_uses_gimg(L, W, PP, QQ);
                const bool lds = !gimg && W == 1 && scan_image_fits_lds(L, W, PP, QQ);
                auto add = [&](bool q, bool g, bool f) {
                    snprintf(buf, sizeof(buf), "em_scan_kernel<%d, %d, %d, %d, %s, %s, %s>\n", PP, QQ, L, W,
                             q ? "true" : "false", g ? "true" : "false", f ? "true" : "false");
                    out += buf;
                };
                if (gimg) { add(true, true, false); add(false, true, true); }
                if (lds) { add(false, false, false); add(true, false, false); add(false, false, true); }
            }
    };
    for (int L : Ls1) scan_names(L, 1);
    for (int W : {2, 4})
        for (int L : LsW) scan_names(L, W);
    for (int lpc : {32, 16})
        for (int L = (lpc == 16 ? 5 : 3); L <= 32; L++)
            for (int PP : widths)
                for (int QQ : widths) {
                    if (!pair_member_fits(L, lpc, PP, QQ)) continue;
                    const bool wide = PP > 4 || QQ > 4;
                    auto add = [&](bool q, bool lead) {
                        snprintf(buf, sizeof(buf), "em_pair_kernel<%d, %d, %d, %d, %s, %s>\n", PP, QQ, L, lpc,
                                 q ? "true" : "false", lead ? "true" : "false");
                        out += buf;
                    };
                    if (!wide) { add(false, false); add(true, false); }
                    if (L <= 16) {                       // LEAD forms: chunks of <= 16 steps
                        if (!wide) add(false, true);
                        add(true, true);                 // (wide inputs: LEAD form, work-queue schedule only)
                    }
                }
}
