// ldsr_api.hip -- the C ABI of include/ldsr_hip.h: argument checks, workspace carving,
// block tables, launches.  No numerics live here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldsr_hip.h"
#include "ldsr_kernels.h"

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(LDSR_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

extern "C" const char *ldsr_last_error(void) { return g_err.c_str(); }
extern "C" const char *ldsr_version(void) { return "ldsr_hip 0.1.0 (gfx950)"; }

extern "C" int ldsr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- optional kernel timer: HIP events around the EM kernel, on its launch stream ------------
static struct {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
} g_prof;

extern "C" void ldsr_profile_enable(int on) {
    g_prof.on = on != 0;
    g_prof.used = 0;
}

extern "C" int ldsr_profile_collect(double *total_ms, int *n_launches) {
    double tot = 0.0;
    for (size_t i = 0; i < g_prof.used; i++) {
        HIPCHK(hipEventSynchronize(g_prof.ev[i].second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[i].first, g_prof.ev[i].second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = (int)g_prof.used;
    g_prof.used = 0;
    return LDSR_OK;
}

static hipError_t prof_mark(hipStream_t stream, bool end) {
    if (!g_prof.on) return hipSuccess;
    if (!end) {
        if (g_prof.used == g_prof.ev.size()) {
            hipEvent_t a, b;
            hipError_t e = hipEventCreate(&a);
            if (e != hipSuccess) return e;
            e = hipEventCreate(&b);
            if (e != hipSuccess) return e;
            g_prof.ev.emplace_back(a, b);
        }
        return hipEventRecord(g_prof.ev[g_prof.used].first, stream);
    }
    return hipEventRecord(g_prof.ev[g_prof.used++].second, stream);
}

extern "C" void ldsr_shutdown(void) {
    for (auto &p : g_prof.ev) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    g_prof.ev.clear();
    g_prof.used = 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int resolve_algo(int algo, int T, int PP, int QQ) {
    if (algo == LDSR_ALGO_AUTO) return em_scan_supported(T, PP, QQ) ? LDSR_ALGO_SCAN : LDSR_ALGO_SERIAL;
    return algo;
}

static int cells_per_block(int algo, int T, int PP, int QQ) {
    return algo == LDSR_ALGO_SCAN ? em_scan_waves_per_block(T, PP, QQ) : 64;
}

struct WsLayout {
    size_t sc, yp, yz, up, vp, blk, soc, queue, scratch, total;
    long scratch_stride;
    int max_blocks;
};

static WsLayout ws_layout(int n_series, int T, int PP, int QQ, int shared_uv, int n_cells,
                          int algo, int cpb) {
    WsLayout L;
    size_t o = 0;
    L.sc = o; o = align256(o + sizeof(SeriesConst) * (size_t)n_series);
    L.yp = o; o = align256(o + sizeof(double) * (size_t)n_series * T);
    L.yz = o; o = align256(o + sizeof(double) * (size_t)n_series * T);
    const size_t nuv = shared_uv ? 1 : (size_t)n_series;
    L.up = o; o = align256(o + sizeof(double) * nuv * T * PP);
    L.vp = o; o = align256(o + sizeof(double) * nuv * T * QQ);
    L.max_blocks = n_cells / cpb + n_series + 1;
    L.blk = o; o = align256(o + sizeof(int) * 3 * (size_t)L.max_blocks);
    L.soc = o; o = align256(o + sizeof(int) * (size_t)(n_cells > 0 ? n_cells : 1));
    L.queue = o; o = align256(o + sizeof(int) * (size_t)n_series);
    L.scratch_stride = ((long)n_cells + 63) / 64 * 64;
    L.scratch = o;
    if (algo == LDSR_ALGO_SERIAL) o = align256(o + sizeof(double) * 2 * (size_t)T * L.scratch_stride);
    L.total = o;
    return L;
}

static int check_common(int n_series, int T, int p, int q, const double *y,
                        const int *cell_offsets) {
    if (n_series < 1) return fail(LDSR_EINVAL, "n_series must be >= 1");
    if (T < 2) return fail(LDSR_EINVAL, "T must be >= 2");
    if (p < 1 || q < 1) return fail(LDSR_EINVAL, "p and q must be >= 1 (use 1 with u/v = NULL for an absent input)");
    if (p > LDSR_MAXPQ || q > LDSR_MAXPQ)
        return fail(LDSR_EUNSUPPORTED, "p and q above 16 are not supported by this build");
    if (!y || !cell_offsets) return fail(LDSR_EINVAL, "y and cell_offsets must not be NULL");
    if (cell_offsets[0] != 0) return fail(LDSR_EINVAL, "cell_offsets[0] must be 0");
    for (int s = 0; s < n_series; s++)
        if (cell_offsets[s + 1] < cell_offsets[s])
            return fail(LDSR_EINVAL, "cell_offsets must be non-decreasing");
    return LDSR_OK;
}

extern "C" size_t ldsr_em_workspace_bytes(int n_series, int T, int p, int q, int n_cells,
                                          int algo) {
    if (n_series < 1 || T < 2 || p < 1 || q < 1 || p > LDSR_MAXPQ || q > LDSR_MAXPQ || n_cells < 0)
        return 0;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    algo = resolve_algo(algo, T, PP, QQ);
    if (algo == LDSR_ALGO_SCAN && !em_scan_supported(T, PP, QQ)) return 0;
    // the layout for shared_uv = 0 is an upper bound for shared_uv = 1
    return ws_layout(n_series, T, PP, QQ, 0, n_cells, algo, cells_per_block(algo, T, PP, QQ)).total;
}

// Runs series_prep on `stream` and fills the workspace pointers.
static int prepare_series(hipStream_t stream, int n_series, int T, int p, int q, int PP, int QQ,
                          const double *d_y, const double *d_u, const double *d_v, int shared_uv,
                          char *ws, const WsLayout &L) {
    PrepParams pp;
    pp.T = T; pp.p = p; pp.q = q; pp.PP = PP; pp.QQ = QQ; pp.shared_uv = shared_uv;
    pp.y = d_y; pp.u = d_u; pp.v = d_v;
    pp.yp = (double *)(ws + L.yp);
    pp.yz = (double *)(ws + L.yz);
    pp.up = (double *)(ws + L.up);
    pp.vp = (double *)(ws + L.vp);
    pp.sc = (SeriesConst *)(ws + L.sc);
    pp.queue = (int *)(ws + L.queue);
    HIPCHK(launch_series_prep(pp, n_series, stream));
    return LDSR_OK;
}

extern "C" int ldsr_em_batch_device(int device, void *stream_, int n_series, int T, int p, int q,
                                    const double *d_y, const double *d_u, const double *d_v,
                                    int shared_uv, const int *cell_offsets,
                                    const double *d_theta0, int niter, double tol, int algo,
                                    double *d_theta, double *d_lik, int *d_n_iter, int *d_status,
                                    double *d_liks, void *d_workspace, size_t workspace_bytes) {
    int rc = check_common(n_series, T, p, q, d_y, cell_offsets);
    if (rc) return rc;
    if (niter < 2) return fail(LDSR_EINVAL, "niter must be >= 2 (the reference reads lik[1], src/EM.cpp:256)");
    if (!(tol >= 0.0)) return fail(LDSR_EINVAL, "tol must be >= 0");
    if (!d_theta0 || !d_theta || !d_lik || !d_n_iter || !d_status || !d_workspace)
        return fail(LDSR_EINVAL, "NULL output / workspace pointer");
    const int n_cells = cell_offsets[n_series];
    if (n_cells == 0) return LDSR_OK;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    algo = resolve_algo(algo, T, PP, QQ);
    if (algo != LDSR_ALGO_SERIAL && algo != LDSR_ALGO_SCAN) return fail(LDSR_EINVAL, "unknown algo");
    if (algo == LDSR_ALGO_SCAN && !em_scan_supported(T, PP, QQ))
        return fail(LDSR_EINVAL, "LDSR_ALGO_SCAN needs T <= 2048 and p, q <= 8");
    const int cpb = cells_per_block(algo, T, PP, QQ);
    const WsLayout L = ws_layout(n_series, T, PP, QQ, shared_uv, n_cells, algo, cpb);
    if (workspace_bytes < L.total)
        return fail(LDSR_EINVAL, "workspace too small: need " + std::to_string(L.total) + " bytes");
    if (((size_t)d_workspace & 255) != 0) return fail(LDSR_EINVAL, "workspace must be 256-byte aligned");
    HIPCHK(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    char *ws = (char *)d_workspace;

    rc = prepare_series(stream, n_series, T, p, q, PP, QQ, d_y, d_u, d_v, shared_uv, ws, L);
    if (rc) return rc;

    // block table: blocks never straddle a series.  Static mapping (serial kernel; scan kernel
    // when tol == 0, i.e. every cell runs exactly niter iterations): (series, first cell, n cells
    // of the block).  Work queue (scan kernel, tol > 0): (series, first cell of the SERIES, n
    // cells of the series) -- waves pull cells from the per-series queue, so a wave whose cell
    // converges early takes the next one instead of idling.
    const bool use_queue = algo == LDSR_ALGO_SCAN && (tol > 0.0 || em_scan_global_image(T, PP, QQ));
    std::vector<int> tab;
    tab.reserve(3 * (size_t)L.max_blocks);
    std::vector<int> bs, bc, bn;
    for (int s = 0; s < n_series; s++)
        for (int c = cell_offsets[s]; c < cell_offsets[s + 1]; c += cpb) {
            bs.push_back(s);
            if (use_queue) {
                bc.push_back(cell_offsets[s]);
                bn.push_back(cell_offsets[s + 1] - cell_offsets[s]);
            } else {
                bc.push_back(c);
                bn.push_back(std::min(cpb, cell_offsets[s + 1] - c));
            }
        }
    const int n_blocks = (int)bs.size();
    if (n_blocks > L.max_blocks) return fail(LDSR_EINVAL, "internal: block table overflow");
    tab.insert(tab.end(), bs.begin(), bs.end());
    tab.insert(tab.end(), bc.begin(), bc.end());
    tab.insert(tab.end(), bn.begin(), bn.end());
    int *d_tab = (int *)(ws + L.blk);
    HIPCHK(hipMemcpyAsync(d_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, stream));

    EmParams prm;
    prm.T = T; prm.p = p; prm.q = q; prm.has_u = d_u != nullptr; prm.has_v = d_v != nullptr;
    prm.niter = niter; prm.n_cells = n_cells; prm.tol = tol;
    prm.yp = (const double *)(ws + L.yp);
    prm.yz = (const double *)(ws + L.yz);
    prm.up = (const double *)(ws + L.up);
    prm.vp = (const double *)(ws + L.vp);
    prm.u_stride = shared_uv ? 0 : (long)T * PP;
    prm.v_stride = shared_uv ? 0 : (long)T * QQ;
    prm.sc = (const SeriesConst *)(ws + L.sc);
    prm.blk_series = d_tab;
    prm.blk_cell0 = d_tab + n_blocks;
    prm.blk_ncell = d_tab + 2 * n_blocks;
    prm.theta0 = d_theta0;
    prm.theta = d_theta; prm.lik = d_lik; prm.liks = d_liks;
    prm.n_iter = d_n_iter; prm.status = d_status;
    prm.queue = (int *)(ws + L.queue);
    prm.scratch = (double *)(ws + L.scratch);
    prm.scratch_stride = L.scratch_stride;
    HIPCHK(prof_mark(stream, false));
    if (algo == LDSR_ALGO_SCAN)
        HIPCHK(launch_em_scan(prm, PP, QQ, n_blocks, cpb, use_queue, stream));
    else
        HIPCHK(launch_em_serial(prm, PP, QQ, n_blocks, stream));
    HIPCHK(prof_mark(stream, true));
    return LDSR_OK;
}

// RAII holder for temporary device buffers of the host-pointer entry points.
struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename Tp>
    hipError_t alloc(Tp **out, size_t n) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, n ? n * sizeof(Tp) : sizeof(Tp));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (Tp *)p;
        return e;
    }
};

struct HostInputs {
    double *d_y = nullptr, *d_u = nullptr, *d_v = nullptr;
};

static int upload_inputs(DevBufs &B, HostInputs &H, int n_series, int T, int p, int q,
                         const double *y, const double *u, const double *v, int shared_uv) {
    const size_t nuv = shared_uv ? 1 : (size_t)n_series;
    HIPCHK(B.alloc(&H.d_y, (size_t)n_series * T));
    HIPCHK(hipMemcpy(H.d_y, y, sizeof(double) * (size_t)n_series * T, hipMemcpyHostToDevice));
    if (u) {
        HIPCHK(B.alloc(&H.d_u, nuv * T * p));
        HIPCHK(hipMemcpy(H.d_u, u, sizeof(double) * nuv * T * p, hipMemcpyHostToDevice));
    }
    if (v) {
        HIPCHK(B.alloc(&H.d_v, nuv * T * q));
        HIPCHK(hipMemcpy(H.d_v, v, sizeof(double) * nuv * T * q, hipMemcpyHostToDevice));
    }
    return LDSR_OK;
}

extern "C" int ldsr_em_batch(int device, int n_series, int T, int p, int q, const double *y,
                             const double *u, const double *v, int shared_uv,
                             const int *cell_offsets, const double *theta0, int niter, double tol,
                             int algo, double *theta, double *lik, int *n_iter, int *status,
                             double *liks) {
    int rc = check_common(n_series, T, p, q, y, cell_offsets);
    if (rc) return rc;
    if (niter < 2) return fail(LDSR_EINVAL, "niter must be >= 2 (the reference reads lik[1], src/EM.cpp:256)");
    if (!theta0 || !theta || !lik || !n_iter || !status) return fail(LDSR_EINVAL, "NULL pointer");
    const int n_cells = cell_offsets[n_series];
    if (n_cells == 0) return LDSR_OK;
    const int P = 6 + p + q;
    HIPCHK(hipSetDevice(device));
    DevBufs B;
    HostInputs H;
    rc = upload_inputs(B, H, n_series, T, p, q, y, u, v, shared_uv);
    if (rc) return rc;
    double *d_theta0, *d_theta, *d_lik, *d_liks = nullptr;
    int *d_n_iter, *d_status;
    char *d_ws;
    HIPCHK(B.alloc(&d_theta0, (size_t)n_cells * P));
    HIPCHK(B.alloc(&d_theta, (size_t)n_cells * P));
    HIPCHK(B.alloc(&d_lik, (size_t)n_cells));
    HIPCHK(B.alloc(&d_n_iter, (size_t)n_cells));
    HIPCHK(B.alloc(&d_status, (size_t)n_cells));
    if (liks) HIPCHK(B.alloc(&d_liks, (size_t)n_cells * niter));
    const size_t wsb = ldsr_em_workspace_bytes(n_series, T, p, q, n_cells, algo);
    HIPCHK(B.alloc(&d_ws, wsb));
    HIPCHK(hipMemcpy(d_theta0, theta0, sizeof(double) * (size_t)n_cells * P, hipMemcpyHostToDevice));
    rc = ldsr_em_batch_device(device, nullptr, n_series, T, p, q, H.d_y, H.d_u, H.d_v, shared_uv,
                              cell_offsets, d_theta0, niter, tol, algo, d_theta, d_lik, d_n_iter,
                              d_status, d_liks, d_ws, wsb);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipMemcpy(theta, d_theta, sizeof(double) * (size_t)n_cells * P, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lik, d_lik, sizeof(double) * (size_t)n_cells, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n_iter, d_n_iter, sizeof(int) * (size_t)n_cells, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, d_status, sizeof(int) * (size_t)n_cells, hipMemcpyDeviceToHost));
    if (liks)
        HIPCHK(hipMemcpy(liks, d_liks, sizeof(double) * (size_t)n_cells * niter, hipMemcpyDeviceToHost));
    return LDSR_OK;
}

// Multi-GPU form of ldsr_em_batch: the cell grid is cut into contiguous slices, one host thread
// per listed device runs its slice through ldsr_em_batch (cells are grouped by series, so a slice
// is a contiguous range of series: plain pointer offsets, no gather).  No collective; results
// land directly in the caller's arrays.  The same device may be listed more than once.
extern "C" int ldsr_em_batch_multi(int n_devices, const int *devices, int n_series, int T, int p,
                                   int q, const double *y, const double *u, const double *v,
                                   int shared_uv, const int *cell_offsets, const double *theta0,
                                   int niter, double tol, int algo, double *theta, double *lik,
                                   int *n_iter, int *status, double *liks) {
    if (n_devices < 1 || !devices) return fail(LDSR_EINVAL, "n_devices must be >= 1");
    int rc = check_common(n_series, T, p, q, y, cell_offsets);
    if (rc) return rc;
    if (niter < 2) return fail(LDSR_EINVAL, "niter must be >= 2 (the reference reads lik[1], src/EM.cpp:256)");
    if (!theta0 || !theta || !lik || !n_iter || !status) return fail(LDSR_EINVAL, "NULL pointer");
    const int n_cells = cell_offsets[n_series];
    const int P = 6 + p + q;
    std::vector<int> rcs((size_t)n_devices, LDSR_OK);
    std::vector<std::string> msgs((size_t)n_devices);
    std::vector<std::thread> pool;
    for (int d = 0; d < n_devices; d++) {
        const int lo = (int)((long long)n_cells * d / n_devices);
        const int hi = (int)((long long)n_cells * (d + 1) / n_devices);
        if (hi <= lo) continue;
        pool.emplace_back([=, &rcs, &msgs]() {
            // series range [s0, s1) that owns cells [lo, hi), and the clipped local offsets
            int s0 = 0;
            while (cell_offsets[s0 + 1] <= lo) s0++;
            int s1 = s0;
            while (s1 < n_series && cell_offsets[s1] < hi) s1++;
            std::vector<int> off((size_t)(s1 - s0) + 1);
            for (int s = s0; s <= s1; s++) {
                int c = cell_offsets[s];
                c = c < lo ? lo : (c > hi ? hi : c);
                off[(size_t)(s - s0)] = c - lo;
            }
            const size_t uo = shared_uv ? 0 : (size_t)s0 * T * p, vo = shared_uv ? 0 : (size_t)s0 * T * q;
            const int r = ldsr_em_batch(devices[d], s1 - s0, T, p, q, y + (size_t)s0 * T,
                                        u ? u + uo : nullptr, v ? v + vo : nullptr, shared_uv,
                                        off.data(), theta0 + (size_t)lo * P, niter, tol, algo,
                                        theta + (size_t)lo * P, lik + lo, n_iter + lo, status + lo,
                                        liks ? liks + (size_t)lo * niter : nullptr);
            rcs[(size_t)d] = r;
            if (r) msgs[(size_t)d] = ldsr_last_error();   // thread-local message of this worker
        });
    }
    for (auto &t : pool) t.join();
    for (int d = 0; d < n_devices; d++)
        if (rcs[(size_t)d])
            return fail(rcs[(size_t)d], "device " + std::to_string(devices[d]) + ": " + msgs[(size_t)d]);
    return LDSR_OK;
}

// Shared driver of the smoother / propagate / mstep host entry points.
// mode: 0 smoother, 1 propagate, 2 mstep, 3 penalized likelihood (smoother, scalar output only)
static int run_fit_kernel(int mode, int device, int n_series, int T, int p, int q, const double *y,
                          const double *u, const double *v, int shared_uv,
                          const int *cell_offsets, const double *theta_in, int stdlik, double *X,
                          double *Y, double *V, double *J, double *lik, double *theta_out,
                          int *status, double lambda = 0.0) {
    int rc = check_common(n_series, T, p, q, y, cell_offsets);
    if (rc) return rc;
    const int n_cells = cell_offsets[n_series];
    if (n_cells == 0) return LDSR_OK;
    const int P = 6 + p + q;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    HIPCHK(hipSetDevice(device));
    DevBufs B;
    HostInputs H;
    rc = upload_inputs(B, H, n_series, T, p, q, y, u, v, shared_uv);
    if (rc) return rc;
    const WsLayout L = ws_layout(n_series, T, PP, QQ, shared_uv, n_cells, LDSR_ALGO_SCAN, 64);
    char *ws;
    HIPCHK(B.alloc(&ws, L.total));
    rc = prepare_series(nullptr, n_series, T, p, q, PP, QQ, H.d_y, H.d_u, H.d_v, shared_uv, ws, L);
    if (rc) return rc;
    std::vector<int> soc((size_t)n_cells);
    for (int s = 0; s < n_series; s++)
        for (int c = cell_offsets[s]; c < cell_offsets[s + 1]; c++) soc[c] = s;
    int *d_soc = (int *)(ws + L.soc);
    HIPCHK(hipMemcpy(d_soc, soc.data(), sizeof(int) * (size_t)n_cells, hipMemcpyHostToDevice));

    SmoothParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.T = T; sp.p = p; sp.q = q; sp.has_u = u != nullptr; sp.has_v = v != nullptr;
    sp.n_cells = n_cells; sp.stdlik = stdlik; sp.mode = (mode == 3) ? 0 : mode;
    sp.lambda = lambda;
    sp.yp = (const double *)(ws + L.yp);
    sp.up = (const double *)(ws + L.up);
    sp.vp = (const double *)(ws + L.vp);
    sp.u_stride = shared_uv ? 0 : (long)T * PP;
    sp.v_stride = shared_uv ? 0 : (long)T * QQ;
    sp.sc = (const SeriesConst *)(ws + L.sc);
    sp.series_of_cell = d_soc;
    const size_t nT = (size_t)n_cells * T;
    double *d_X, *d_Y, *d_V, *d_J, *d_lik, *d_theta;
    int *d_status;
    HIPCHK(B.alloc(&d_X, nT));
    HIPCHK(B.alloc(&d_Y, nT));
    HIPCHK(B.alloc(&d_V, nT));
    HIPCHK(B.alloc(&d_J, nT));
    HIPCHK(B.alloc(&d_lik, (size_t)n_cells));
    HIPCHK(B.alloc(&d_theta, (size_t)n_cells * P));
    HIPCHK(B.alloc(&d_status, (size_t)n_cells));
    sp.X = d_X; sp.Y = d_Y; sp.V = d_V; sp.J = d_J; sp.lik = d_lik;
    sp.status = d_status;
    if (mode == 2) {
        if (!X || !V || !J || !theta_out) return fail(LDSR_EINVAL, "mstep needs X, V, J and theta");
        HIPCHK(hipMemcpy(d_X, X, sizeof(double) * nT, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_V, V, sizeof(double) * nT, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_J, J, sizeof(double) * nT, hipMemcpyHostToDevice));
        sp.theta_out = d_theta;
        HIPCHK(launch_mstep(sp, PP, QQ, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
        HIPCHK(hipMemcpy(theta_out, d_theta, sizeof(double) * (size_t)n_cells * P, hipMemcpyDeviceToHost));
        if (status) HIPCHK(hipMemcpy(status, d_status, sizeof(int) * (size_t)n_cells, hipMemcpyDeviceToHost));
        return LDSR_OK;
    }
    if (!theta_in || !lik) return fail(LDSR_EINVAL, "theta and lik must not be NULL");
    HIPCHK(hipMemcpy(d_theta, theta_in, sizeof(double) * (size_t)n_cells * P, hipMemcpyHostToDevice));
    sp.theta = d_theta;
    double *d_pen = nullptr;
    if (mode == 3) {
        HIPCHK(B.alloc(&d_pen, (size_t)n_cells));
        sp.pen = d_pen;
    }
    HIPCHK(launch_smooth(sp, PP, QQ, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    if (mode == 3) {
        HIPCHK(hipMemcpy(lik, d_pen, sizeof(double) * (size_t)n_cells, hipMemcpyDeviceToHost));
        return LDSR_OK;
    }
    if (X) HIPCHK(hipMemcpy(X, d_X, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (Y) HIPCHK(hipMemcpy(Y, d_Y, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (V) HIPCHK(hipMemcpy(V, d_V, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (J && mode == 0) HIPCHK(hipMemcpy(J, d_J, sizeof(double) * nT, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lik, d_lik, sizeof(double) * (size_t)n_cells, hipMemcpyDeviceToHost));
    return LDSR_OK;
}

extern "C" int ldsr_smooth_batch(int device, int n_series, int T, int p, int q, const double *y,
                                 const double *u, const double *v, int shared_uv,
                                 const int *cell_offsets, const double *theta, int stdlik,
                                 double *X, double *Y, double *V, double *J, double *lik) {
    return run_fit_kernel(0, device, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, theta,
                          stdlik, X, Y, V, J, lik, nullptr, nullptr);
}

extern "C" int ldsr_penalized_lik_batch(int device, int n_series, int T, int p, int q,
                                        const double *y, const double *u, const double *v,
                                        int shared_uv, const int *cell_offsets, const double *theta,
                                        double lambda, double *pl) {
    return run_fit_kernel(3, device, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, theta, 0,
                          nullptr, nullptr, nullptr, nullptr, pl, nullptr, nullptr, lambda);
}

extern "C" int ldsr_propagate_batch(int device, int n_series, int T, int p, int q, const double *y,
                                    const double *u, const double *v, int shared_uv,
                                    const int *cell_offsets, const double *theta, int stdlik,
                                    double *X, double *Y, double *V, double *lik) {
    return run_fit_kernel(1, device, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, theta,
                          stdlik, X, Y, V, nullptr, lik, nullptr, nullptr);
}

extern "C" int ldsr_mstep_batch(int device, int n_series, int T, int p, int q, const double *y,
                                const double *u, const double *v, int shared_uv,
                                const int *cell_offsets, const double *X, const double *V,
                                const double *J, double *theta, int *status) {
    return run_fit_kernel(2, device, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, nullptr,
                          1, (double *)X, nullptr, (double *)V, (double *)J, nullptr, theta, status);
}

// R/LDS_reconstruction.R:50-58: best lik among models with C > 0 (NaN ignored); if no
// model has C > 0, which.max(liks).  First index on ties; -1 if nothing is selectable.
extern "C" int ldsr_select_restart(int n, const double *lik, const double *theta, int p, int q) {
    if (n <= 0 || !lik || !theta) return -1;
    const int P = 6 + p + q;
    bool any_pos = false;
    for (int i = 0; i < n; i++)
        if (theta[(size_t)i * P + 1 + p] > 0) { any_pos = true; break; }
    int best = -1;
    for (int i = 0; i < n; i++) {
        if (std::isnan(lik[i])) continue;
        if (any_pos && !(theta[(size_t)i * P + 1 + p] > 0)) continue;
        if (best < 0 || lik[i] > lik[best]) best = i;
    }
    return best;
}
