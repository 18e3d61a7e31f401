// ldsr_api.hip -- the C ABI of include/ldsr_hip.h: argument checks, device arenas, workspace
// carving, block tables, launches, restart selection.  No numerics live here.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldsr_hip.h"
#include "ldsr_kernels.h"
#include "em_pair_impl.h"      // (layout constants only)
#include "source_hash.h"       // LDSR_SOURCE_HASH, written by the Makefile

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
extern "C" const char *ldsr_version(void) { return "ldsr_hip 0.4.0 (gfx950) src " LDSR_SOURCE_HASH; }
extern "C" const char *ldsr_source_hash(void) { return LDSR_SOURCE_HASH; }

extern "C" int ldsr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- user interrupts ---------------------------------------------------------------------------
// One host-pinned flag for the whole library: kernels launched while a callback is registered
// poll it; only the thread that entered the library from outside (t_poll) runs the callback.
// The registration is ONE immutable record published through one atomic pointer (callback and
// argument as two atomics let a poller racing a re-registration call the old callback with the new
// argument); replaced records live until ldsr_shutdown.  An entering call announces itself (`active`)
// BEFORE it looks at the registration, and shutdown withdraws the registration BEFORE it looks at
// `active`: with sequentially consistent atomics either the call sees no registration and never
// touches the flag, or shutdown sees the call and leaves the flag alone.
struct IntrReg {
    int (*cb)(void *);
    void *arg;
};
static struct {
    std::mutex mu;                    // serialises registration and shutdown
    std::atomic<const IntrReg *> reg{nullptr};
    std::atomic<int *> flag{nullptr};  // pinned, portable
    std::atomic<int> active{0};       // external EM calls in flight
    std::vector<const IntrReg *> retired;
} g_intr;
static thread_local bool t_poll = false;     // this thread may run the callback
static thread_local bool t_worker = false;   // a library worker thread: never runs it

extern "C" int ldsr_set_interrupt_callback(int (*callback)(void *), void *arg) {
    std::lock_guard<std::mutex> lk(g_intr.mu);
    if (callback && !g_intr.flag.load()) {
        void *p = nullptr;
        HIPCHK(hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped));
        *(int *)p = 0;
        g_intr.flag.store((int *)p);
    }
    const IntrReg *old = g_intr.reg.load();
    if (old && callback && old->cb == callback && old->arg == arg) return LDSR_OK;     // (the shim registers before every call)
    g_intr.reg.store(callback ? new IntrReg{callback, arg} : nullptr);
    if (old) g_intr.retired.push_back(old);
    return LDSR_OK;
}

static const int *intr_flag_for_kernels() { return g_intr.reg.load() ? g_intr.flag.load() : nullptr; }
static bool intr_raised() {
    int *f = g_intr.flag.load();
    return g_intr.reg.load() && f && *(volatile int *)f != 0;
}

// Run the callback (external caller thread only); raise the flag if it asks to stop.
static void intr_poll() {
    const IntrReg *r = g_intr.reg.load();
    int *f = g_intr.flag.load();
    if (!t_poll || !r || !f) return;
    if (*(volatile int *)f == 0 && r->cb(r->arg)) *(volatile int *)f = 1;
}

// RAII around an external EM entry: the outermost call on a non-worker thread becomes the poller
// and clears a stale flag when no other call is in flight.
struct IntrScope {
    bool owner = false;
    IntrScope() {
        if (t_worker || t_poll) return;
        const int before = g_intr.active.fetch_add(1);       // announce first ...
        if (!g_intr.reg.load()) {                             // ... then look (see above)
            g_intr.active.fetch_sub(1);
            return;
        }
        owner = true;
        t_poll = true;
        int *f = g_intr.flag.load();
        if (before == 0 && f) *(volatile int *)f = 0;
    }
    ~IntrScope() {
        if (!owner) return;
        // the last external call to leave clears the flag: a raised interrupt stops every call that
        // is in flight and is over once they have all returned
        int *f = g_intr.flag.load();
        if (g_intr.active.fetch_sub(1) == 1 && f) *(volatile int *)f = 0;
        t_poll = false;
    }
};

// Wait for a stream; the polling thread keeps the interrupt callback alive meanwhile.
static hipError_t wait_stream(hipStream_t stream) {
    if (!t_poll) return hipStreamSynchronize(stream);
    for (unsigned n = 0;; n++) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if ((n & 15) == 15) intr_poll();
        usleep(50);
    }
}

// ---- optional kernel timer: HIP events around the EM kernel, on its launch stream ------------
// Slots are handed out under a mutex (ldsr_em_batch_multi / _groups call in from worker
// threads); an event pair belongs to the device it was created on.
struct ProfSlot {
    int device;
    hipEvent_t a, b;
};
static struct {
    std::mutex mu;
    bool on = false;
    std::vector<ProfSlot> ev;
    size_t used = 0;
} g_prof;

extern "C" void ldsr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    g_prof.used = 0;
}

extern "C" int ldsr_profile_collect(double *total_ms, int *n_launches) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot = 0.0;
    for (size_t i = 0; i < g_prof.used; i++) {
        HIPCHK(hipEventSynchronize(g_prof.ev[i].b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[i].a, g_prof.ev[i].b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = (int)g_prof.used;
    g_prof.used = 0;
    return LDSR_OK;
}

// Records the start event and returns the slot (-1 when the timer is off).
static hipError_t prof_begin(int device, hipStream_t stream, int *slot) {
    *slot = -1;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on) return hipSuccess;
    if (g_prof.used == g_prof.ev.size()) {
        ProfSlot s;
        s.device = device;
        hipError_t e = hipEventCreate(&s.a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&s.b);
        if (e != hipSuccess) return e;
        g_prof.ev.push_back(s);
    } else if (g_prof.ev[g_prof.used].device != device) {
        ProfSlot &s = g_prof.ev[g_prof.used];
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
        s.device = device;
        hipError_t e = hipEventCreate(&s.a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&s.b);
        if (e != hipSuccess) return e;
    }
    *slot = (int)g_prof.used++;
    return hipEventRecord(g_prof.ev[*slot].a, stream);
}

static hipError_t prof_end(hipStream_t stream, int slot) {
    if (slot < 0) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    return hipEventRecord(g_prof.ev[slot].b, stream);
}

// ---- pinned staging ring: small host tables that must reach the device asynchronously --------
// ldsr_em_batch_device promises to only enqueue work, so its block table cannot be copied from
// the caller's (or a function-local) pageable memory: it is written into a library-owned pinned
// slot, copied from there on the caller's stream, and the slot is recycled once its event has
// completed (normally long before the ring wraps around).
struct StageSlot {
    int device = -1;
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};
static struct {
    std::mutex mu;
    StageSlot slot[16];
    unsigned next = 0;
} g_stage;

static int stage_h2d_async(int device, hipStream_t stream, void *dst, const void *src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_stage.mu);
    StageSlot &s = g_stage.slot[g_stage.next++ % 16];
    if (s.pending) {
        HIPCHK(hipEventSynchronize(s.ev));
        s.pending = false;
    }
    if (s.device != device && s.ev) {       // events belong to a device
        (void)hipEventDestroy(s.ev);
        s.ev = nullptr;
    }
    if (!s.ev) HIPCHK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    s.device = device;
    if (s.cap < bytes) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr;
        s.cap = 0;
        const size_t cap = std::max(align256(bytes) * 2, (size_t)16384);
        HIPCHK(hipHostMalloc(&s.host, cap, hipHostMallocDefault));
        s.cap = cap;
    }
    memcpy(s.host, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, s.host, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(s.ev, stream));
    s.pending = true;
    return LDSR_OK;
}

// ---- device arenas of the host-pointer entry points -------------------------------------------
// One arena = one device block + one pinned host block (both grow-only) + one non-blocking
// stream.  A call leases a free arena of its device (or creates one), so concurrent callers never
// share buffers, and repeated calls of the same shape do no hipMalloc / hipFree at all.
struct Arena {
    int device = -1;
    hipStream_t stream = nullptr;
    char *dev = nullptr, *pin = nullptr;
    size_t dev_cap = 0, pin_cap = 0;
    bool busy = false;
};
static std::mutex g_arena_mu;
static std::vector<Arena *> g_arenas;

static int arena_acquire(int device, Arena **out) {
    HIPCHK(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_arena_mu);
    for (Arena *a : g_arenas)
        if (a->device == device && !a->busy) {
            a->busy = true;
            *out = a;
            return LDSR_OK;
        }
    Arena *a = new Arena;
    a->device = device;
    hipError_t e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete a;
        return fail(LDSR_EHIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
    }
    a->busy = true;
    g_arenas.push_back(a);
    *out = a;
    return LDSR_OK;
}

static void arena_release(Arena *a) {
    if (!a) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    a->busy = false;
}

struct ArenaLease {     // releases on scope exit
    Arena *a = nullptr;
    ~ArenaLease() { arena_release(a); }
};

static int arena_reserve(Arena *a, size_t dev_bytes, size_t pin_bytes) {
    if (dev_bytes > a->dev_cap) {
        HIPCHK(hipStreamSynchronize(a->stream));
        if (a->dev) (void)hipFree(a->dev);
        a->dev = nullptr;
        a->dev_cap = 0;
        const size_t cap = align256(dev_bytes + dev_bytes / 8);
        HIPCHK(hipMalloc((void **)&a->dev, cap));
        a->dev_cap = cap;
    }
    if (pin_bytes > a->pin_cap) {
        HIPCHK(hipStreamSynchronize(a->stream));
        if (a->pin) (void)hipHostFree(a->pin);
        a->pin = nullptr;
        a->pin_cap = 0;
        const size_t cap = align256(pin_bytes + pin_bytes / 8);
        HIPCHK(hipHostMalloc((void **)&a->pin, cap, hipHostMallocDefault));
        a->pin_cap = cap;
    }
    return LDSR_OK;
}

extern "C" void ldsr_shutdown(void) {
    {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        for (auto &p : g_prof.ev) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        g_prof.ev.clear();
        g_prof.used = 0;
    }
    {
        std::lock_guard<std::mutex> lk(g_stage.mu);
        for (StageSlot &s : g_stage.slot) {
            if (s.pending) (void)hipEventSynchronize(s.ev);
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.host) (void)hipHostFree(s.host);
            s = StageSlot();
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_intr.mu);
        const IntrReg *cur = g_intr.reg.exchange(nullptr);     // withdraw first, then look at `active`
        if (g_intr.active.load() == 0) {
            if (g_intr.flag.load()) (void)hipHostFree(g_intr.flag.load());
            g_intr.flag.store(nullptr);
            for (const IntrReg *r : g_intr.retired) delete r;
            g_intr.retired.clear();
            delete cur;
        } else if (cur) {
            g_intr.retired.push_back(cur);      // calls in flight: the flag and the records stay
        }
    }
    std::lock_guard<std::mutex> lk(g_arena_mu);
    std::vector<Arena *> keep;
    for (Arena *a : g_arenas) {
        if (a->busy) {          // a call is still running on another thread: leave it alone
            keep.push_back(a);
            continue;
        }
        if (hipSetDevice(a->device) == hipSuccess) {
            (void)hipStreamSynchronize(a->stream);
            if (a->dev) (void)hipFree(a->dev);
            if (a->pin) (void)hipHostFree(a->pin);
            (void)hipStreamDestroy(a->stream);
        }
        delete a;
    }
    g_arenas.swap(keep);
}

// bump allocator over an arena block: first pass sizes, second pass hands out pointers
struct Carver {
    size_t o = 0;
    size_t take(size_t bytes) {
        const size_t r = o;
        o = align256(o + (bytes ? bytes : 8));
        return r;
    }
};

// ---- algorithm choice / workspace layout -------------------------------------------------------
// compute units of a device (cached)
static int device_cu_count(int device) {
    static std::mutex mu;
    static std::vector<int> cache;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)cache.size() <= device) cache.resize((size_t)device + 1, 0);
    if (!cache[(size_t)device]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        cache[(size_t)device] = n;
    }
    return cache[(size_t)device];
}

// LDSR_PAIR=0 in the environment keeps AUTO off the two-cells-per-wave kernel (same-box A/B runs)
// LDSR_FORCE_FILL=1: AUTO treats every launch as large enough for the pair family (tests and the
// fuzzer exercise AUTO's choices with a handful of cells)
static bool force_fill() {
    static const bool on = [] { const char *e = getenv("LDSR_FORCE_FILL"); return e && e[0] == '1'; }();
    return on;
}

// name of the EM kernel of the most recent launch per device (ldsr_last_em_kernel)
static std::mutex g_last_mu;
static std::vector<std::string> g_last_kernel;
static void remember_kernel(int device, const char *name) {
    std::lock_guard<std::mutex> lk(g_last_mu);
    if ((int)g_last_kernel.size() <= device) g_last_kernel.resize((size_t)device + 1);
    g_last_kernel[(size_t)device] = name;
}
extern "C" int ldsr_last_em_kernel(int device, char *buf, size_t len) {
    std::lock_guard<std::mutex> lk(g_last_mu);
    if (device < 0 || (int)g_last_kernel.size() <= device || g_last_kernel[(size_t)device].empty() || !buf || !len)
        return -1;
    snprintf(buf, len, "%s", g_last_kernel[(size_t)device].c_str());
    return 0;
}

// LDSR_LEAD=0 keeps AUTO off the closed-form lead of the pair family (same-box A/B runs)
static bool lead_enabled() {
    static const bool on = [] { const char *e = getenv("LDSR_LEAD"); return !(e && e[0] == '0'); }();
    return on;
}

// LDSR_STEADY_ORDER=0: the steady form takes the cells in the caller's order (A/B runs)
static bool order_enabled() {
    static const bool on = [] { const char *e = getenv("LDSR_STEADY_ORDER"); return !(e && e[0] == '0'); }();
    return on;
}

static bool pair_enabled() {
    static const bool on = [] { const char *e = getenv("LDSR_PAIR"); return !(e && e[0] == '0'); }();
    return on;
}

// AUTO resolves to LDSR_ALGO_PAIR as the name of the several-cells-per-wave family (two or four
// cells per wave); which member -- or the scan kernel after all -- runs is decided per launch
// (em_batch_device_impl: launch size, tol, fully observed or not).
static int resolve_algo(int algo, int T, int PP, int QQ) {
    if (algo == LDSR_ALGO_AUTO) {
        if (pair_enabled() && (em_pair_supported(T, PP, QQ, 32) || em_pair_supported(T, PP, QQ, 16)))
            return LDSR_ALGO_PAIR;
        return em_scan_supported(T, PP, QQ) ? LDSR_ALGO_SCAN : LDSR_ALGO_SERIAL;
    }
    return algo;
}

// cells per workgroup of the EM launch (the workspace's block table is sized for the scan
// kernel's value, the smallest of them, whenever its image is built)
static int cells_per_block(int algo, int T, int PP, int QQ, int lpc = 32, int lead = 0) {
    if (algo == LDSR_ALGO_PAIR || algo == LDSR_ALGO_QUAD) {
        if (algo == LDSR_ALGO_QUAD) lpc = 16;
        const int c = em_pair_cells_per_block(T, PP, QQ, lpc, lead);
        return c > 0 ? c : 16;
    }
    return algo == LDSR_ALGO_SCAN ? em_scan_cells_per_block(T, PP, QQ) : 64;
}

struct WsLayout {
    size_t sc, yp, yz, up, vp, img, img2, img3, blk, soc, queue, perm, perm_key, scratch, total;
    long scratch_stride, img_stride;   // img_stride: doubles per series image (0 = no image)
    long img2_stride;                  // pair kernel's image (0 = none)
    long img3_stride;                  // ... and its lead image
    int max_blocks, img_L, img_NL, img2_L, img2_NL = 32, lead = 0;
    // cell order of the pair kernel's steady form (series_prep orders when order_on): set per launch
    bool order_on = false;
    bool build_scan_img = true;        // false: a pair-family launch with no smoother pass behind it
    int order_cpb = 0, order_ntr = 0;
    const double *order_theta0 = nullptr;
    const int *order_off = nullptr;     // device copy of the cell offsets (behind the block table)
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
    // chunk-transposed images for the scan kernel (also built for serial-kernel EM launches
    // whose shape the scan kernel supports: the winners' fit then runs on the scan kernel)
    em_scan_layout(T, PP, QQ, &L.img_L, &L.img_NL, &L.img_stride);
    L.img = o; o = align256(o + sizeof(double) * (size_t)L.img_stride * n_series);
    // room for the pair family's images whatever runs in the end (the launch decides: member,
    // chunk length, a closed-form lead): the largest 32-lane image, and u_t of a lead of up to T steps
    L.img2_stride = 0; L.img2_L = 0; L.img3_stride = 0;
    if (PP <= 8 && QQ <= 8 && algo != LDSR_ALGO_SERIAL) {
        // (wide inputs: the LEAD form's tail only, chunks of <= 16 steps)
        L.img2_stride = pair_image_doubles(PP <= 4 && QQ <= 4 ? 32 : 16, PP, QQ, 32);
        L.img3_stride = (long)(T + 32) * PP;
    }
    L.img2 = o; o = align256(o + sizeof(double) * (size_t)L.img2_stride * n_series);
    L.img3 = o; o = align256(o + sizeof(double) * (size_t)L.img3_stride * n_series);
    if (L.img_stride) cpb = std::min(cpb, em_scan_cells_per_block(T, PP, QQ));   // the winners' FIT launch
    if (PP <= 8 && QQ <= 8 && algo != LDSR_ALGO_SERIAL) cpb = std::min(cpb, 4);    // the pair family's smallest workgroup
    L.max_blocks = n_cells / cpb + n_series + 1;
    // (block table, then the device copy of the cell offsets for series_prep's cell ordering)
    L.blk = o; o = align256(o + sizeof(int) * (3 * (size_t)L.max_blocks + (size_t)n_series + 1));
    L.soc = o; o = align256(o + sizeof(int) * (size_t)(n_cells > 0 ? n_cells : 1));
    L.queue = o; o = align256(o + sizeof(int) * (size_t)n_series);
    // cell order of the pair kernel's steady form (two cells per wave, narrow inputs): position -> cell, and the keys
    const bool may_order = PP <= 4 && QQ <= 4 && algo != LDSR_ALGO_SERIAL;
    L.perm = o; o = align256(o + (may_order ? sizeof(int) * (size_t)(n_cells > 0 ? n_cells : 1) : 0));
    L.perm_key = o; o = align256(o + (may_order ? sizeof(int) * (size_t)(n_cells > 0 ? n_cells : 1) : 0));
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

static int check_em(int niter, double tol) {
    if (niter < 2) return fail(LDSR_EINVAL, "niter must be >= 2 (the reference reads lik[1], src/EM.cpp:256)");
    if (!(tol >= 0.0)) return fail(LDSR_EINVAL, "tol must be >= 0");
    return LDSR_OK;
}

extern "C" size_t ldsr_em_workspace_bytes(int n_series, int T, int p, int q, int n_cells,
                                          int algo) {
    if (n_series < 1 || T < 2 || p < 1 || q < 1 || p > LDSR_MAXPQ || q > LDSR_MAXPQ || n_cells < 0)
        return 0;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    const int algo_in = algo;
    algo = resolve_algo(algo, T, PP, QQ);
    if (algo == LDSR_ALGO_SCAN && !em_scan_supported(T, PP, QQ)) return 0;
    if (algo == LDSR_ALGO_PAIR && !em_pair_supported(T, PP, QQ, 32) && !(algo_in == LDSR_ALGO_AUTO && em_pair_supported(T, PP, QQ, 16))) return 0;
    if (algo == LDSR_ALGO_QUAD && !em_pair_supported(T, PP, QQ, 16)) return 0;
    // the layout for shared_uv = 0 is an upper bound for shared_uv = 1
    return ws_layout(n_series, T, PP, QQ, 0, n_cells, algo, cells_per_block(algo, T, PP, QQ)).total;
}

// scan kernel: cells converge at their own pace (tol > 0) -> per-series work queue
static bool scan_uses_queue(int T, int PP, int QQ, double tol) {
    return tol > 0.0 || em_scan_queue_only(T, PP, QQ);
}

// the tail [T - tail, T) AUTO sweeps when the first lead_steps steps of every series are missing
// (0: no closed-form lead) -- the same rule as em_batch_device_impl
// early stopping on series that may have missing steps: does the two-cells-per-wave kernel still
// beat the scan kernel?  (measured: only in four-wave workgroups with chunks of <= 13 steps)
static bool pair_pays_with_early_stopping(int T, int PP, int QQ) {
    return T <= 416 && em_pair_waves_per_block(T, PP, QQ, 32, 0) == 4;
}

// Runs to convergence on fully observed series with wide-ish inputs (padded p + q >= 8): since the scan kernel reads its
// image ahead (round 4) it beats the two-cells-per-wave kernel on long chunks (T = 813 (3,3) 20 000 cells 2.46 against
// 2.79 ms), and four cells per wave lose to two (T = 260 (4,4) 20 000 cells 1.13 against 0.97): tools/auto_regret.py,
// profiles/r04_auto_regret.txt.
static bool conv_wide(double tol, int PP, int QQ) { return tol > 0.0 && PP + QQ >= 8; }

// does the LEAD form at lp lanes per cell fit a CU's LDS: the tail's image, the strips and the lead's u_t
static bool lead_fits(int T, int tail, int PP, int QQ, int lp) {
    const bool wide = PP > 4 || QQ > 4;
    int Lc = 0;
    long img = 0;
    em_pair_layout(tail, PP, QQ, lp, &Lc, &img, true);
    if (!img) return false;
    const size_t lds = ((size_t)img + (wide ? 4 : 8) * (size_t)pair_strip_doubles(Lc) +
                        (size_t)pair_lead_doubles(T - tail, lp, PP)) * sizeof(double);
    return lds <= 160 * 1024;
}
// four cells per wave for the tail of a lead: narrow inputs, tails of <= 256 steps (p = 3, 4 since the
// lead's second pass sums 5 + 2 p values instead of 7 + 4 p: they fit the 16-lane reduction now)
static bool lead_quad(int T, int tail, int PP, int QQ) {
    return PP <= 8 && QQ <= 8 && tail <= 256 && em_pair_supported(tail, PP, QQ, 16, true) && lead_fits(T, tail, PP, QQ, 16);
}

static bool lead_short34(int T, int tail, int PP) { return PP > 2 && T - tail < 512; }

static int lead_tail(int T, int PP, int QQ, int lead_steps) {
    if (!pair_enabled() || !lead_enabled() || lead_steps < 192 || PP > 8 || QQ > 8) return 0;
    const bool wide = PP > 4 || QQ > 4;     // (two cells per wave, LEAD form only: kernels_scan.hip pair_plan)
    int tail = std::max(T - lead_steps, 80);
    tail = (tail + 15) / 16 * 16;
    static const int max_tail = [] { const char *e = getenv("LDSR_LEAD_MAX_TAIL"); return e ? atoi(e) : 512; }();
    if (tail > max_tail || tail > 512 || T - tail < 128) return 0;
    // p = 3, 4 have the two-cells-per-wave LEAD form only (7 + 4 p lead sums per step): on short
    // series the four-cells-per-wave kernel over all T steps is quicker (tools/auto_regret.py, same
    // box: T = 260 (4,4) 20 000 cells 1.86 ms against 1.29, to convergence 11.4 against 7.0; the Nakhon
    // Phanom shape, T = 813 with a lead of 733 steps, keeps it: 3.29 -> 2.50 ms)
    // (with four cells per wave -- p = 3, 4 since round 3, launches of >= 3/4 of a round -- the lead pays on
    // short series too: the caller checks lead_short34() before it falls back to two cells per wave)
    if (PP > 2 && T - tail < 512 && !lead_quad(T, tail, PP, QQ)) return 0;
    // the lead's u_t live in LDS behind the tail's image and the strips
    (void)wide;
    if (lead_quad(T, tail, PP, QQ)) return tail;
    // (padded p = 8 up to T = 1024: four cells per wave or the scan kernel, see em_batch_device_impl)
    return ((PP < 8 || T > 1024) && lead_fits(T, tail, PP, QQ, 32)) ? tail : 0;
}

static int em_plan_impl(int T, int p, int q, int niter, double tol, int algo, char *buf, size_t len,
                        bool fully_observed);

extern "C" int ldsr_em_plan_lead(int T, int p, int q, int niter, double tol, int algo, int lead_steps,
                                 char *buf, size_t len) {
    if (T < 2 || p < 1 || q < 1 || p > LDSR_MAXPQ || q > LDSR_MAXPQ || niter < 2 || !(tol >= 0.0))
        return -1;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    const int tail = algo == LDSR_ALGO_AUTO ? lead_tail(T, PP, QQ, lead_steps) : 0;
    if (!tail) return em_plan_impl(T, p, q, niter, tol, algo, buf, len, lead_steps < 0);
    const int lpc = lead_quad(T, tail, PP, QQ) ? 16 : 32;
    if (buf && len) em_pair_kernel_name(tail, PP, QQ, lpc, tol > 0.0 || PP > 4 || QQ > 4, buf, len, true);
    return lpc == 16 ? LDSR_ALGO_QUAD : LDSR_ALGO_PAIR;
}

// the kernel one Kalman_smoother pass of this shape runs (launch_smoother): the FIT form of the scan
// kernel where its plan holds, else the serial smoother (returns LDSR_ALGO_SERIAL, empty name)
extern "C" int ldsr_smooth_plan(int T, int p, int q, char *buf, size_t len) {
    if (T < 2 || p < 1 || q < 1 || p > LDSR_MAXPQ || q > LDSR_MAXPQ) return -1;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    if (buf && len) buf[0] = 0;
    if (!em_scan_supported(T, PP, QQ)) return LDSR_ALGO_SERIAL;
    if (buf && len) em_scan_kernel_name(T, PP, QQ, false, true, buf, len);
    return LDSR_ALGO_SCAN;
}

// names of every compiled instantiation of the scan and pair families, one per line; returns the
// length needed (including the terminating 0)
extern "C" size_t ldsr_kernel_inventory(char *buf, size_t len) {
    std::string s;
    em_kernel_inventory(s);
    if (buf && len) snprintf(buf, len, "%s", s.c_str());
    return s.size() + 1;
}

extern "C" int ldsr_em_plan(int T, int p, int q, int niter, double tol, int algo, char *buf,
                            size_t len) {
    return em_plan_impl(T, p, q, niter, tol, algo, buf, len, false);
}

static int em_plan_impl(int T, int p, int q, int niter, double tol, int algo, char *buf, size_t len,
                        bool fully_observed) {
    if (T < 2 || p < 1 || q < 1 || p > LDSR_MAXPQ || q > LDSR_MAXPQ || niter < 2 || !(tol >= 0.0))
        return -1;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    const bool was_auto = algo == LDSR_ALGO_AUTO;
    algo = resolve_algo(algo, T, PP, QQ);
    // what ldsr_em_batch_device runs (the host-pointer entries additionally take the pair / quad
    // kernels with tol > 0 when every series is fully observed)
    const bool masked_conv = was_auto && algo == LDSR_ALGO_PAIR && tol > 0.0 && !fully_observed;
    if (masked_conv && !pair_pays_with_early_stopping(T, PP, QQ)) algo = LDSR_ALGO_SCAN;
    if (was_auto && algo == LDSR_ALGO_PAIR && conv_wide(tol, PP, QQ) && T > 512 && em_scan_supported(T, PP, QQ)) algo = LDSR_ALGO_SCAN;
    if (algo == LDSR_ALGO_SCAN) {
        if (!em_scan_supported(T, PP, QQ)) return -1;
        if (buf && len) em_scan_kernel_name(T, PP, QQ, scan_uses_queue(T, PP, QQ, tol), false, buf, len);
    } else if (algo == LDSR_ALGO_PAIR || algo == LDSR_ALGO_QUAD) {
        // AUTO (a launch that fills the device assumed): four cells per wave where they fit, else two
        int lpc = algo == LDSR_ALGO_QUAD ? 16 : 32;
        if (was_auto && !conv_wide(tol, PP, QQ) && em_pair_supported(T, PP, QQ, 16)) { lpc = 16; algo = LDSR_ALGO_QUAD; }
        if (!em_pair_supported(T, PP, QQ, lpc)) return -1;
        if (buf && len) em_pair_kernel_name(T, PP, QQ, lpc, tol > 0.0, buf, len);
    } else if (algo == LDSR_ALGO_SERIAL) {
        if (buf && len) em_serial_kernel_name(T, PP, QQ, buf, len);
    } else {
        return -1;
    }
    return algo;
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
    pp.img = (L.img_stride && L.build_scan_img) ? (double *)(ws + L.img) : nullptr;
    pp.img_stride = L.img_stride;
    pp.L = L.img_L;
    pp.NL = L.img_NL;
    pp.img2 = L.img2_stride ? (double *)(ws + L.img2) : nullptr;
    pp.img2_stride = L.img2_stride;
    pp.L2 = L.img2_L;
    pp.NL2 = L.img2_NL;
    pp.lead = L.lead;
    pp.img3 = (L.lead > 0 && L.img3_stride) ? (double *)(ws + L.img3) : nullptr;
    pp.img3_stride = L.img3_stride;
    pp.n_series = n_series;
    pp.perm = L.order_on ? (int *)(ws + L.perm) : nullptr;
    pp.perm_key = (int *)(ws + L.perm_key);
    pp.cell_off = L.order_off;
    pp.theta0 = L.order_theta0;
    pp.order_cpb = L.order_cpb;
    pp.order_ntr = L.order_ntr;
    HIPCHK(launch_series_prep(pp, n_series, stream));
    return LDSR_OK;
}

// liks_nanfill: entries of d_liks beyond a cell's n_iter are set to NaN (the ABI contract of the
// batch entry points); the restart-grid path reads only the first n_iter entries and skips it.
static int em_batch_device_impl(int device, hipStream_t stream, int n_series, int T, int p, int q,
                                const double *d_y, const double *d_u, const double *d_v,
                                int shared_uv, const int *cell_offsets, const double *d_theta0,
                                int niter, double tol, int algo, double *d_theta, double *d_lik,
                                int *d_n_iter, int *d_status, double *d_liks, int liks_nanfill,
                                void *d_workspace, size_t workspace_bytes,
                                const int *abort_flag = nullptr, int dense_hint = -1,
                                int *algo_used = nullptr, int lead_hint = -1, int *lead_used = nullptr,
                                int lead_force = 0, const int *plan_off = nullptr, int plan_ns = 0,
                                bool fit_follows = true) {
    int rc = check_common(n_series, T, p, q, d_y, cell_offsets);
    if (rc) return rc;
    rc = check_em(niter, tol);
    if (rc) return rc;
    if (!d_theta0 || !d_theta || !d_lik || !d_n_iter || !d_status || !d_workspace)
        return fail(LDSR_EINVAL, "NULL output / workspace pointer");
    const int n_cells = cell_offsets[n_series];
    if (n_cells == 0) return LDSR_OK;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    const bool was_auto = algo == LDSR_ALGO_AUTO;
    algo = resolve_algo(algo, T, PP, QQ);
    const int algo_layout = algo;       // what the workspace was sized and laid out for
    // Is the launch large enough for the pair family to pay?  Counted in CUs' worth of cells (eight
    // waves).  Eight-wave workgroups (long chunks): >= 7/8 of the CUs.  Four-wave
    // workgroups (short series, two per CU): the shared per-wave work pays much earlier --
    // same box, T = 400 (1,2) / 200 (2,2) / 300 (1,4), scan -> pair -> quad in ms: 1024 cells
    // 0.47 -> 0.39 -> 0.54, 2048 0.50 -> 0.42 -> 0.57, 3072 0.74 -> 0.62 -> 0.59, 4096 0.94 -> 0.68 ->
    // 0.62 (tools/fill_ab.sh) -- two cells per wave from 1/4 of the CUs (1024 cells), four from 3/8
    // (3072 cells).
    auto fills = [&](int Te, int lp, bool lead_form = false) {
        if (!em_pair_supported(Te, PP, QQ, lp, lead_form)) return false;
        const int c = (64 / lp) * 8;          // a CU's eight waves
        long wgs = 0;
        // (a slice of a multi-device call counts the whole call's cells: plan_off)
        const int *po = plan_off ? plan_off : cell_offsets;
        const int pn = plan_off ? plan_ns : n_series;
        for (int s = 0; s < pn; s++) wgs += (po[s + 1] - po[s] + c - 1) / c;
        const long cus = device_cu_count(device);
        if (force_fill()) return true;
        // (runs to convergence: only up to chunks of 13 steps -- beyond, the scan kernel with its read-ahead wins on
        // launches of this size: T = 600 (1,2) 2000 cells 0.284 against 0.313 ms, (2,4) 0.384 against 0.512)
        if (!lead_form && em_pair_waves_per_block(Te, PP, QQ, lp, 0) == 4 && (tol == 0.0 || Te <= 416))
            return wgs * 8 >= (lp == 16 ? 3 : 2) * cus;
        // the closed-form lead skips most of the work, so it pays from ~1536 cells (same box, scan ->
        // LEAD in ms, tools/lead_fill_ab.sh: T = 2000 (1,4) 1536 cells 1.94 -> 1.40, 3072 3.79 -> 1.47;
        // T = 4000 (2,2) 1536 cells 8.27 -> 2.47; T = 813 (3,3) 1536 0.92 -> 0.85, 3072 1.42 -> 1.27);
        // with early stopping only for long leads (2048 cells: T = 2000 2.81 -> 2.23, T = 813 2.85 -> 3.09)
        // from T = 1536 on the scan kernel's chunks are 28..32 steps long (and its image may live in global
        // memory): there the lead pays whatever the launch size -- 50 lone cells, niter = 200, scan -> LEAD in
        // ms: T = 2000 (1,4) 3.61 -> 2.44, (3,5) 17.1 -> 3.96, T = 4000 (2,2) 6.30 -> 4.95; at T = 1100..1300
        // it is a toss-up (1.40 -> 1.48, 1.81 -> 2.07, 2.17 -> 1.78)
        // (four cells per wave with p = 3, 4 or wide inputs -- possible since the lead's second pass sums
        // 5 + 2 p values -- pay from 3072 cells, with leads of 1024 steps and more from 6144: same box, two ->
        // four cells per wave in ms, T = 813 (3,3) 8192 cells 2.33 -> 1.55, 4096 1.24 -> 1.06, 3072 1.21 -> 1.03,
        // 2048 0.82 -> 0.98; (4,8) T = 1024 3072 cells 1.52 -> 1.26, 2048 1.06 -> 1.22; T = 2000 (3,4) 4096
        // cells 2.25 -> 2.64, 6144 4.33 -> 2.73)
        if (lead_form && lp == 16 && (PP > 2 || QQ > 4)) return wgs * (lead_hint >= 1024 ? 4 : 8) >= 3 * cus;
        if (lead_form && T >= 1536) return true;
        if (lead_form && (tol == 0.0 || lead_hint >= 1024))
            return wgs * (lp == 16 ? 16 : 8) >= 3 * cus;
        return wgs * 8 >= 7 * cus;
    };
    // A long all-missing lead common to every series (paleo-type data; lead_hint from the caller
    // that has seen y): the pair family's LEAD form handles it in closed form and sweeps only the
    // tail -- [T - tail, T) with tail a multiple of 16 of at most 512 steps (chunks of <= 16 steps:
    // four cells per wave up to 256 steps, two beyond).
    int lead = 0, lpc = algo == LDSR_ALGO_QUAD ? 16 : 32;
    if (was_auto && algo != LDSR_ALGO_SERIAL) {
        const int tail = lead_tail(T, PP, QQ, lead_hint);
        if (tail) {
            if (lead_quad(T, tail, PP, QQ) && fills(tail, 16, true)) { lead = T - tail; lpc = 16; algo = LDSR_ALGO_QUAD; }
            // (two cells per wave with padded p = 8 run ONE four-wave workgroup per CU and lose to the scan kernel:
            // T = 813 (7,7) 2048 cells 1.76 against 1.40 ms, 4096 cells 3.03 against 2.66, four per wave 1.89)
            // (beyond T = 1024 the scan kernel's chunks are long and they win again)
            else if ((PP < 8 || T > 1024) && !lead_short34(T, tail, PP) && fills(tail, 32, true)) { lead = T - tail; lpc = 32; algo = LDSR_ALGO_PAIR; }
        }
    }
    if (lead_force > 0 && (algo == LDSR_ALGO_PAIR || algo == LDSR_ALGO_QUAD)) lead = lead_force;   // (a re-run of part of a batch)
    if (lead_used) *lead_used = lead;
    const int Te = T - lead;            // steps the sweeps of the pair family work on
    if (!lead) {
        // AUTO with early stopping: the pair kernel couples two cells per wave and sixteen per
        // workgroup (= per CU), so widely different iteration counts cost it more than they cost the
        // scan kernel's four-cell workgroups.  Measured (converged runs, tol = 1e-5): fully observed
        // series (cells stop after 28..63 iterations) pair +8..12 %; masked series (4..176, cfg5 up
        // to 745 iterations) pair -2..-24 %.  So with tol > 0 AUTO takes the pair kernel only for
        // series known to be fully observed (the host-pointer entries look; dense_hint).
        // (Short series are the exception: in the two-cells-per-wave kernel's four-wave workgroups --
        // chunks of <= 13 steps -- the coupling costs less than the shared per-wave work saves:
        // T = 300 (2,2) +34 %, T = 400 (1,2) +11..33 %; from T = 500 on the scan kernel wins by 5..24 %.)
        const bool masked_conv = was_auto && algo == LDSR_ALGO_PAIR && tol > 0.0 && dense_hint != 1;
        if (masked_conv && !pair_pays_with_early_stopping(T, PP, QQ)) algo = LDSR_ALGO_SCAN;
        if (was_auto && algo == LDSR_ALGO_PAIR && conv_wide(tol, PP, QQ) && T > 512 && em_scan_supported(T, PP, QQ)) algo = LDSR_ALGO_SCAN;
        // ... and only when its workgroups (one per CU: 16 cells at two cells per wave, 32 at four) fill
        // the device: 512 cells are 32 pair workgroups on 32 of 256 CUs but 128 scan workgroups on 128
        // of them (a quarter of the time).  Four cells per wave where they fit and fill, else two.
        if (was_auto && algo == LDSR_ALGO_PAIR) {
            // (masked series with early stopping reach this point only as short series: there four
            // cells per wave win once the launch is large -- tools/auto_regret.py, 20 000 cells:
            // T = 260 (4,4) 9.5 -> 7.0 ms, T = 150 (1,2) 2.96 -> 2.74; at 2000 cells two per wave stay ahead)
            if (!conv_wide(tol, PP, QQ) && fills(T, 16)) { lpc = 16; algo = LDSR_ALGO_QUAD; }
            else if (fills(T, 32)) lpc = 32;
            else algo = em_scan_supported(T, PP, QQ) ? LDSR_ALGO_SCAN : LDSR_ALGO_SERIAL;
        }
    }
    if (algo != LDSR_ALGO_SERIAL && algo != LDSR_ALGO_SCAN && algo != LDSR_ALGO_PAIR && algo != LDSR_ALGO_QUAD)
        return fail(LDSR_EINVAL, "unknown algo");
    if (algo == LDSR_ALGO_SCAN && !em_scan_supported(T, PP, QQ))
        return fail(LDSR_EINVAL, "LDSR_ALGO_SCAN needs T <= 8192 and p, q <= 8 (and T >= L (L - 1) for its chunk length)");
    if (algo == LDSR_ALGO_PAIR && !em_pair_supported(Te, PP, QQ, 32, lead > 0))
        return fail(LDSR_EINVAL, "LDSR_ALGO_PAIR needs 65 <= T <= 1024, p, q <= 4 and a series image that leaves room for eight waves per CU (ldsr_em_plan tells)");
    if (algo == LDSR_ALGO_QUAD && !em_pair_supported(Te, PP, QQ, 16, lead > 0))
        return fail(LDSR_EINVAL, "LDSR_ALGO_QUAD needs 65 <= T <= 512, p, q <= 4 (ldsr_em_plan tells)");
    const bool cpw = algo == LDSR_ALGO_PAIR || algo == LDSR_ALGO_QUAD;                // the pair family's body
    if (algo_used) *algo_used = algo;
    const int cpb = cells_per_block(algo, cpw ? Te : T, PP, QQ, lpc, cpw ? lead : 0);
    WsLayout L = ws_layout(n_series, T, PP, QQ, shared_uv, n_cells, algo_layout,
                           cells_per_block(algo_layout, T, PP, QQ));
    // (the scan kernel's image serves the winners' FIT pass of the restart-grid entries; the bare device
    // entries run no smoother behind a pair-family launch: series_prep skips it -- 4096 values at config 2)
    L.build_scan_img = fit_follows || !cpw;
    if (cpw) {       // the image of the member that runs (the room is for the largest)
        long sz = 0;
        em_pair_layout(Te, PP, QQ, lpc, &L.img2_L, &sz, lead > 0);
        L.img2_NL = lpc;
        L.lead = lead;
    } else {
        L.img2_stride = 0;       // no pair-family launch: series_prep skips its images
        L.img3_stride = 0;
    }
    if (workspace_bytes < L.total)
        return fail(LDSR_EINVAL, "workspace too small: need " + std::to_string(L.total) + " bytes");
    if (((size_t)d_workspace & 255) != 0) return fail(LDSR_EINVAL, "workspace must be 256-byte aligned");
    HIPCHK(hipSetDevice(device));
    char *ws = (char *)d_workspace;

    // block table: blocks never straddle a series.  Static mapping (serial kernel; scan kernel
    // when tol == 0, i.e. every cell runs exactly niter iterations): (series, first cell, n cells
    // of the block).  Work queue (scan kernel, tol > 0): (series, first cell of the SERIES, n
    // cells of the series) -- waves pull cells from the per-series queue, so a wave whose cell
    // converges early takes the next one instead of idling.
    const bool use_queue = (algo == LDSR_ALGO_SCAN && scan_uses_queue(T, PP, QQ, tol)) ||
                           (cpw && (tol > 0.0 || (lead > 0 && (PP > 4 || QQ > 4))));   // (wide LEAD forms: work-queue schedule only)
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
    std::vector<int> tab;
    tab.reserve(3 * (size_t)n_blocks);
    tab.insert(tab.end(), bs.begin(), bs.end());
    tab.insert(tab.end(), bc.begin(), bc.end());
    tab.insert(tab.end(), bn.begin(), bn.end());
    int *d_tab = (int *)(ws + L.blk);
    // Steady form of the two-cells-per-wave kernel (fully observed series, chunks of >= 24 steps):
    // series_prep also orders every series' cells by predicted slowness (em_pair_impl.h
    // em_pair_body_steady); it reads the cell offsets from the device copy behind the block table.
    const bool steady_launch = cpw && lpc == 32 && lead == 0 && pair_steady(L.img2_L, 32, PP, QQ) && order_enabled();
    if (steady_launch) {
        tab.insert(tab.end(), cell_offsets, cell_offsets + n_series + 1);
        L.order_on = true;
        L.order_cpb = use_queue ? 0 : cpb;
        L.order_ntr = L.img2_L - 1;
        L.order_theta0 = d_theta0;
        L.order_off = d_tab + 3 * n_blocks;
    }
    rc = stage_h2d_async(device, stream, d_tab, tab.data(), sizeof(int) * tab.size());
    if (rc) return rc;
    rc = prepare_series(stream, n_series, T, p, q, PP, QQ, d_y, d_u, d_v, shared_uv, ws, L);
    if (rc) return rc;

    EmParams prm;
    prm.T = T; prm.p = p; prm.q = q; prm.has_u = d_u != nullptr; prm.has_v = d_v != nullptr;
    prm.niter = niter; prm.n_cells = n_cells; prm.tol = tol;
    prm.liks_nanfill = liks_nanfill;
    prm.abort = abort_flag;
    prm.yp = (const double *)(ws + L.yp);
    prm.yz = (const double *)(ws + L.yz);
    prm.up = (const double *)(ws + L.up);
    prm.vp = (const double *)(ws + L.vp);
    prm.u_stride = shared_uv ? 0 : (long)T * PP;
    prm.v_stride = shared_uv ? 0 : (long)T * QQ;
    prm.img = (const double *)(ws + L.img);
    prm.img_stride = L.img_stride;
    prm.img2 = L.img2_stride ? (const double *)(ws + L.img2) : nullptr;
    prm.img2_stride = L.img2_stride;
    prm.lead = cpw ? lead : 0;
    prm.img3 = (cpw && lead > 0) ? (const double *)(ws + L.img3) : nullptr;
    prm.img3_stride = L.img3_stride;
    prm.fitX = prm.fitY = prm.fitV = prm.fitJ = prm.pen = nullptr;
    prm.lambda = 0.0;
    prm.stdlik = 1;
    prm.sc = (const SeriesConst *)(ws + L.sc);
    prm.blk_series = d_tab;
    prm.blk_cell0 = d_tab + n_blocks;
    prm.blk_ncell = d_tab + 2 * n_blocks;
    prm.theta0 = d_theta0;
    prm.theta = d_theta; prm.lik = d_lik; prm.liks = d_liks;
    prm.n_iter = d_n_iter; prm.status = d_status;
    prm.queue = (int *)(ws + L.queue);
    prm.perm = steady_launch ? (const int *)(ws + L.perm) : nullptr;
    prm.scratch = (double *)(ws + L.scratch);
    prm.scratch_stride = L.scratch_stride;
    int slot;
    HIPCHK(prof_begin(device, stream, &slot));
    {
        char nm[160];
        if (cpw) em_pair_kernel_name(Te, PP, QQ, lpc, use_queue, nm, sizeof(nm), lead > 0);
        else if (algo == LDSR_ALGO_SCAN) em_scan_kernel_name(T, PP, QQ, use_queue, false, nm, sizeof(nm));
        else em_serial_kernel_name(T, PP, QQ, nm, sizeof(nm));
        remember_kernel(device, nm);
    }
    if (cpw)
        HIPCHK(launch_em_pair(prm, PP, QQ, lpc, n_blocks, use_queue, stream));
    else if (algo == LDSR_ALGO_SCAN)
        HIPCHK(launch_em_scan(prm, PP, QQ, n_blocks, use_queue, false, stream));
    else
        HIPCHK(launch_em_serial(prm, PP, QQ, n_blocks, stream));
    HIPCHK(prof_end(stream, slot));
    return LDSR_OK;
}

extern "C" int ldsr_em_batch_device_lead(int device, void *stream_, int n_series, int T, int p, int q,
                                         const double *d_y, const double *d_u, const double *d_v,
                                         int shared_uv, const int *cell_offsets,
                                         const double *d_theta0, int niter, double tol, int algo,
                                         double *d_theta, double *d_lik, int *d_n_iter, int *d_status,
                                         double *d_liks, void *d_workspace, size_t workspace_bytes,
                                         int lead_steps) {
    return em_batch_device_impl(device, (hipStream_t)stream_, n_series, T, p, q, d_y, d_u, d_v,
                                shared_uv, cell_offsets, d_theta0, niter, tol, algo, d_theta, d_lik,
                                d_n_iter, d_status, d_liks, 1, d_workspace, workspace_bytes, nullptr,
                                lead_steps < 0 ? 1 : -1, nullptr, lead_steps < 0 ? 0 : lead_steps, nullptr, 0,
                                nullptr, 0, false);
}

extern "C" int ldsr_em_batch_device(int device, void *stream_, int n_series, int T, int p, int q,
                                    const double *d_y, const double *d_u, const double *d_v,
                                    int shared_uv, const int *cell_offsets,
                                    const double *d_theta0, int niter, double tol, int algo,
                                    double *d_theta, double *d_lik, int *d_n_iter, int *d_status,
                                    double *d_liks, void *d_workspace, size_t workspace_bytes) {
    return em_batch_device_impl(device, (hipStream_t)stream_, n_series, T, p, q, d_y, d_u, d_v,
                                shared_uv, cell_offsets, d_theta0, niter, tol, algo, d_theta, d_lik,
                                d_n_iter, d_status, d_liks, 1, d_workspace, workspace_bytes, nullptr, -1,
                                nullptr, -1, nullptr, 0, nullptr, 0, false);
}

// One Kalman_smoother pass (src/EM.cpp:22-131) for n cells on prepared series: the FIT form of
// the scan kernel (one wave group per cell) whenever the shape is supported, else the serial
// one-thread-per-cell kernel.  blk_*: host block table rows (series, first cell, cells) of the
// scan launch; it travels through the pinned staging ring.  d_tab: device room for 3*n_blocks
// ints.  mode: 0 smoother, 1 propagate (serial kernel only).
static int launch_smoother(int device, hipStream_t stream, int T, int p, int q, int PP, int QQ,
                           bool has_u, bool has_v, int shared_uv, char *ws, const WsLayout &L,
                           int n, const std::vector<int> &series_of_cell, const double *d_theta,
                           int stdlik, int mode, double lambda, double *d_X, double *d_Y,
                           double *d_V, double *d_J, double *d_lik, double *d_pen, int *d_status,
                           int *d_soc, bool scalar_only, const int *d_tab_prebuilt = nullptr) {
    if (mode == 0 && L.img_stride && em_scan_supported(T, PP, QQ)) {
        const int cpb = em_scan_cells_per_block(T, PP, QQ);
        const int *d_tab = d_tab_prebuilt;      // [3][n]: one block per cell, filled on the device
        int n_blocks = n;
        if (!d_tab) {
            std::vector<int> bs, bc, bn;
            for (int c = 0; c < n;) {               // blocks never straddle a series
                int e = c + 1;
                while (e < n && e - c < cpb && series_of_cell[(size_t)e] == series_of_cell[(size_t)c]) e++;
                bs.push_back(series_of_cell[(size_t)c]);
                bc.push_back(c);
                bn.push_back(e - c);
                c = e;
            }
            n_blocks = (int)bs.size();
            if (n_blocks > L.max_blocks) return fail(LDSR_EINVAL, "internal: block table overflow (fit)");
            std::vector<int> tab;
            tab.insert(tab.end(), bs.begin(), bs.end());
            tab.insert(tab.end(), bc.begin(), bc.end());
            tab.insert(tab.end(), bn.begin(), bn.end());
            int *d_ws_tab = (int *)(ws + L.blk);
            int rc = stage_h2d_async(device, stream, d_ws_tab, tab.data(), sizeof(int) * tab.size());
            if (rc) return rc;
            d_tab = d_ws_tab;
        }
        EmParams prm;
        memset(&prm, 0, sizeof(prm));
        prm.T = T; prm.p = p; prm.q = q; prm.has_u = has_u; prm.has_v = has_v;
        prm.niter = 1; prm.n_cells = n; prm.tol = 0.0;
        prm.yp = (const double *)(ws + L.yp);
        prm.yz = (const double *)(ws + L.yz);
        prm.up = (const double *)(ws + L.up);
        prm.vp = (const double *)(ws + L.vp);
        prm.u_stride = shared_uv ? 0 : (long)T * PP;
        prm.v_stride = shared_uv ? 0 : (long)T * QQ;
        prm.img = (const double *)(ws + L.img);
        prm.img_stride = L.img_stride;
        prm.sc = (const SeriesConst *)(ws + L.sc);
        prm.blk_series = d_tab;
        prm.blk_cell0 = d_tab + n_blocks;
        prm.blk_ncell = d_tab + 2 * n_blocks;
        prm.queue = (int *)(ws + L.queue);
        prm.theta0 = d_theta;
        prm.lik = d_lik;
        prm.status = d_status;
        prm.fitX = scalar_only ? nullptr : d_X; prm.fitY = scalar_only ? nullptr : d_Y;
        prm.fitV = scalar_only ? nullptr : d_V; prm.fitJ = scalar_only ? nullptr : d_J;
        prm.pen = d_pen;
        prm.lambda = lambda;
        prm.stdlik = stdlik;
        HIPCHK(launch_em_scan(prm, PP, QQ, n_blocks, false, true, stream));
        return LDSR_OK;
    }
    int rc = stage_h2d_async(device, stream, d_soc, series_of_cell.data(), sizeof(int) * (size_t)n);
    if (rc) return rc;
    SmoothParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.T = T; sp.p = p; sp.q = q; sp.has_u = has_u; sp.has_v = has_v;
    sp.n_cells = n; sp.stdlik = stdlik; sp.mode = mode;
    sp.lambda = lambda;
    sp.yp = (const double *)(ws + L.yp);
    sp.up = (const double *)(ws + L.up);
    sp.vp = (const double *)(ws + L.vp);
    sp.u_stride = shared_uv ? 0 : (long)T * PP;
    sp.v_stride = shared_uv ? 0 : (long)T * QQ;
    sp.sc = (const SeriesConst *)(ws + L.sc);
    sp.series_of_cell = d_soc;
    sp.theta = d_theta;
    sp.X = d_X; sp.Y = d_Y; sp.V = d_V; sp.J = d_J;
    sp.lik = d_lik;
    sp.status = d_status;
    sp.pen = d_pen;
    sp.scalar_only = scalar_only;
    HIPCHK(launch_smooth(sp, PP, QQ, stream));
    return LDSR_OK;
}

// ---- one contiguous slice of the cell grid on one device ---------------------------------------
// Phase 1 (slice_run): ONE pinned->device copy of [y | u | v | theta0], series_prep + EM kernel,
// ONE device->pinned copy of [theta | lik | n_iter | status].  Phase 2 (slice_fit_winners, the
// restart-grid entry only): gather the winners' theta and likelihood traces on the device, one
// smoother pass for them, one copy back.  The arena stays leased between the phases.
struct Slice {
    // inputs (host pointers already offset to the slice; `off` are local cell offsets)
    int device = 0, n_series = 0, T = 0, p = 0, q = 0, shared_uv = 0, niter = 0, algo = 0;
    int dense_hint = -1;     // 1: every y_t of every series is finite (AUTO's kernel choice with tol > 0)
    int algo_used = 0;       // the algorithm the batch launch resolved to
    int lead_used = 0;       // ... and the closed-form lead it used
    int lead_hint = -1;      // steps before the first observation of any series
    double tol = 0.0;
    const double *y = nullptr, *u = nullptr, *v = nullptr, *theta0 = nullptr;
    std::vector<int> off;
    // Striped cut (make_slices): the slice holds EVERY series of the call and, of series s, the
    // cells [g_lo[s], g_lo[s] + off[s+1] - off[s]) of the caller's arrays -- theta0 / theta / lik /
    // n_iter / status / liks are the caller's whole arrays, gathered and scattered per series.
    std::vector<int> g_lo;
    // AUTO looks at the whole call, not at the slice: the kernel (hence the rounding of the results)
    // must not depend on how many devices share the work
    const int *plan_off = nullptr;
    int plan_ns = 0;
    // host outputs of phase 1 (liks optional: the full [n_cells][niter] trace, NaN padded)
    double *theta = nullptr, *lik = nullptr, *liks = nullptr;
    int *n_iter = nullptr, *status = nullptr;
    int max_winners = 0;        // > 0: reserve phase-2 buffers and keep the traces on the device
    // Fused restart path (the whole grid on this one slice): selection, winner gather and the
    // winners' fit are enqueued right behind the EM kernel and everything comes back with ONE
    // synchronisation; the per-cell arrays cross PCIe only if the caller asked for them.
    bool fuse = false, want_all = true;
    int *h_winner = nullptr, *h_nit_w = nullptr;
    double *h_theta_w = nullptr, *h_lik_w = nullptr, *h_liks_w = nullptr;
    double *h_X = nullptr, *h_Y = nullptr, *h_V = nullptr, *h_J = nullptr;
    bool fused_done = false;
    // state
    ArenaLease lease;
    int n_cells = 0, PP = 0, QQ = 0, P = 0;
    bool trace_on_device = false;
    WsLayout L;
    size_t wsb = 0;
    size_t d_in = 0, d_y = 0, d_u = 0, d_v = 0, d_th0 = 0, d_off = 0, in_bytes = 0;      // device offsets
    size_t d_out = 0, d_theta = 0, d_lik = 0, d_nit = 0, d_st = 0, out_bytes = 0;
    size_t d_liks = 0, d_ws = 0;
    size_t d_w = 0, w_bytes = 0;          // phase-2 device block
    size_t p_in = 0, p_out = 0, p_w = 0;  // pinned offsets
};

static size_t liks_trace_cap_bytes() {
    const char *e = getenv("LDSR_LIKS_TRACE_MAX_BYTES");
    if (e && *e) return (size_t)strtoull(e, nullptr, 10);
    return (size_t)8 << 30;
}

// phase-2 block layout for n winners (offsets relative to its start)
struct WinLayout {
    size_t cell, ser, theta, liks, X, Y, V, J, lik, st, theta0, blk, lik_w, nit_w, total;
    size_t out_begin, out_bytes;    // [cell | theta | liks | X | Y | V | J | lik | lik_w | nit_w] is copied back
};
static WinLayout win_layout(int n, int P, int T, int niter) {
    WinLayout W;
    Carver c;
    W.ser = c.take(sizeof(int) * n);
    W.theta0 = c.take(sizeof(double) * n * P);
    W.st = c.take(sizeof(int) * n);
    W.blk = c.take(sizeof(int) * 3 * n);
    W.out_begin = c.o;
    W.cell = c.take(sizeof(int) * n);        // winners' cell indices (uploaded, or selected on the device)
    W.theta = c.take(sizeof(double) * n * P);
    W.liks = c.take(sizeof(double) * (size_t)n * niter);
    W.X = c.take(sizeof(double) * (size_t)n * T);
    W.Y = c.take(sizeof(double) * (size_t)n * T);
    W.V = c.take(sizeof(double) * (size_t)n * T);
    W.J = c.take(sizeof(double) * (size_t)n * T);
    W.lik = c.take(sizeof(double) * n);
    W.lik_w = c.take(sizeof(double) * n);
    W.nit_w = c.take(sizeof(int) * n);
    W.out_bytes = c.o - W.out_begin;
    W.total = c.o;
    return W;
}

// per-cell results of the slice (pinned block `pout`) -> the caller's arrays, series by series
static void slice_scatter(const Slice &S, const char *pout) {
    const int P = S.P;
    for (int s = 0; s < S.n_series; s++) {
        const size_t lo = (size_t)S.off[(size_t)s], nc = (size_t)S.off[(size_t)s + 1] - lo, g = (size_t)S.g_lo[(size_t)s];
        if (!nc) continue;
        memcpy(S.theta + g * P, pout + (S.d_theta - S.d_out) + sizeof(double) * lo * P, sizeof(double) * nc * P);
        memcpy(S.lik + g, pout + (S.d_lik - S.d_out) + sizeof(double) * lo, sizeof(double) * nc);
        memcpy(S.n_iter + g, pout + (S.d_nit - S.d_out) + sizeof(int) * lo, sizeof(int) * nc);
        memcpy(S.status + g, pout + (S.d_st - S.d_out) + sizeof(int) * lo, sizeof(int) * nc);
    }
}

static int slice_run(Slice &S) {
    S.n_cells = S.off[S.n_series];
    S.P = 6 + S.p + S.q;
    S.PP = ldsr_pad_dim(S.p);
    S.QQ = ldsr_pad_dim(S.q);
    if (S.n_cells == 0) return LDSR_OK;
    const int T = S.T, P = S.P, n = S.n_cells;
    const size_t nuv = S.shared_uv ? 1 : (size_t)S.n_series;
    int rc = arena_acquire(S.device, &S.lease.a);
    if (rc) return rc;
    Arena *A = S.lease.a;

    Carver c;
    S.d_in = c.o;
    S.d_y = c.take(sizeof(double) * (size_t)S.n_series * T);
    S.d_u = c.take(S.u ? sizeof(double) * nuv * T * S.p : 0);
    S.d_v = c.take(S.v ? sizeof(double) * nuv * T * S.q : 0);
    S.d_th0 = c.take(sizeof(double) * (size_t)n * P);
    S.d_off = c.take(sizeof(int) * ((size_t)S.n_series + 1));
    S.in_bytes = c.o - S.d_in;
    S.d_out = c.o;
    S.d_theta = c.take(sizeof(double) * (size_t)n * P);
    S.d_lik = c.take(sizeof(double) * (size_t)n);
    S.d_nit = c.take(sizeof(int) * (size_t)n);
    S.d_st = c.take(sizeof(int) * (size_t)n);
    S.out_bytes = c.o - S.d_out;
    const size_t trace_bytes = sizeof(double) * (size_t)n * S.niter;
    S.trace_on_device = S.liks != nullptr || (S.max_winners > 0 && trace_bytes <= liks_trace_cap_bytes());
    S.d_liks = c.take(S.trace_on_device ? trace_bytes : 0);
    S.wsb = ldsr_em_workspace_bytes(S.n_series, T, S.p, S.q, n, S.algo);
    if (!S.wsb) return fail(LDSR_EINVAL, "unsupported (T, p, q, algo) combination");
    {   // the layout em_batch_device_impl will carve out of the workspace (phase 2 reuses it)
        const int a = resolve_algo(S.algo, T, S.PP, S.QQ);
        S.L = ws_layout(S.n_series, T, S.PP, S.QQ, S.shared_uv, n, a, cells_per_block(a, T, S.PP, S.QQ));
    }
    S.d_ws = c.take(S.wsb);
    const WinLayout W = win_layout(std::max(S.max_winners, 1), P, T, S.niter);
    S.d_w = c.take(S.max_winners > 0 ? W.total : 0);
    S.w_bytes = S.max_winners > 0 ? W.total : 0;
    const size_t dev_total = c.o;
    // pinned: inputs, outputs, phase-2 block (same relative layouts as on the device)
    S.p_in = 0;
    S.p_out = align256(S.in_bytes);
    S.p_w = S.p_out + align256(S.out_bytes);
    const size_t pin_total = S.p_w + S.w_bytes;
    rc = arena_reserve(A, dev_total, pin_total);
    if (rc) return rc;

    // stage inputs
    char *pin = A->pin + S.p_in;
    memcpy(pin + (S.d_y - S.d_in), S.y, sizeof(double) * (size_t)S.n_series * T);
    {
        bool all_obs = true;
        const size_t ny = (size_t)S.n_series * T;
        for (size_t i = 0; i < ny && all_obs; i++) all_obs = std::isfinite(S.y[i]);
        S.dense_hint = all_obs ? 1 : 0;
        // all-missing lead common to every series (the pair family's closed form)
        int lead = T;
        for (int s = 0; s < S.n_series && lead > 0; s++) {
            int t = 0;
            while (t < lead && !std::isfinite(S.y[(size_t)s * T + t])) t++;
            lead = std::min(lead, t);
        }
        S.lead_hint = lead;
    }
    if (S.u) memcpy(pin + (S.d_u - S.d_in), S.u, sizeof(double) * nuv * T * S.p);
    if (S.v) memcpy(pin + (S.d_v - S.d_in), S.v, sizeof(double) * nuv * T * S.q);
    for (int s = 0; s < S.n_series; s++)      // this slice's cells of every series
        memcpy(pin + (S.d_th0 - S.d_in) + sizeof(double) * (size_t)S.off[(size_t)s] * P,
               S.theta0 + (size_t)S.g_lo[(size_t)s] * P,
               sizeof(double) * (size_t)(S.off[(size_t)s + 1] - S.off[(size_t)s]) * P);
    memcpy(pin + (S.d_off - S.d_in), S.off.data(), sizeof(int) * ((size_t)S.n_series + 1));
    HIPCHK(hipMemcpyAsync(A->dev + S.d_in, pin, S.in_bytes, hipMemcpyHostToDevice, A->stream));

    rc = em_batch_device_impl(
        S.device, A->stream, S.n_series, T, S.p, S.q, (const double *)(A->dev + S.d_y),
        S.u ? (const double *)(A->dev + S.d_u) : nullptr, S.v ? (const double *)(A->dev + S.d_v) : nullptr,
        S.shared_uv, S.off.data(), (const double *)(A->dev + S.d_th0), S.niter, S.tol, S.algo,
        (double *)(A->dev + S.d_theta), (double *)(A->dev + S.d_lik), (int *)(A->dev + S.d_nit),
        (int *)(A->dev + S.d_st), S.trace_on_device ? (double *)(A->dev + S.d_liks) : nullptr,
        S.liks != nullptr, A->dev + S.d_ws, S.wsb, intr_flag_for_kernels(), S.dense_hint, &S.algo_used,
        S.lead_hint, &S.lead_used, 0, S.plan_off, S.plan_ns);
    if (rc) return rc;
    char *pout = A->pin + S.p_out;
    if (S.fuse && S.trace_on_device) {
        // selection, winner extraction and the winners' fit behind the EM kernel, one sync
        const int ns = S.n_series;
        char *dw = A->dev + S.d_w, *pw = A->pin + S.p_w;
        SelectParams sel;
        sel.n_series = ns; sel.P = P; sel.c_index = 1 + S.p;
        sel.off = (const int *)(A->dev + S.d_off);
        sel.theta = (const double *)(A->dev + S.d_theta);
        sel.lik = (const double *)(A->dev + S.d_lik);
        sel.winner = (int *)(dw + W.cell);
        HIPCHK(launch_select_winners(sel, A->stream));
        GatherParams gp;
        gp.n_w = ns; gp.P = P; gp.niter = S.niter;
        gp.cell = (const int *)(dw + W.cell);
        gp.theta = sel.theta;
        gp.theta0 = (const double *)(A->dev + S.d_th0);
        gp.n_iter = (const int *)(A->dev + S.d_nit);
        gp.liks = (const double *)(A->dev + S.d_liks);
        gp.theta_w = (double *)(dw + W.theta);
        gp.theta0_w = (double *)(dw + W.theta0);
        gp.liks_w = (double *)(dw + W.liks);
        gp.lik = sel.lik;
        gp.lik_w = (double *)(dw + W.lik_w);
        gp.n_iter_w = (int *)(dw + W.nit_w);
        gp.blk = (int *)(dw + W.blk);
        HIPCHK(launch_gather_winners(gp, A->stream));
        const bool want_fit = S.h_X || S.h_Y || S.h_V || S.h_J;
        if (want_fit) {
            std::vector<int> soc((size_t)ns);
            for (int i = 0; i < ns; i++) soc[(size_t)i] = i;
            char *ws = A->dev + S.d_ws;
            rc = launch_smoother(S.device, A->stream, T, S.p, S.q, S.PP, S.QQ, S.u != nullptr,
                                 S.v != nullptr, S.shared_uv, ws, S.L, ns, soc,
                                 (const double *)(dw + W.theta), 1, 0, 0.0, (double *)(dw + W.X),
                                 (double *)(dw + W.Y), (double *)(dw + W.V), (double *)(dw + W.J),
                                 (double *)(dw + W.lik), nullptr, (int *)(dw + W.st),
                                 (int *)(dw + W.ser), false, (const int *)(dw + W.blk));
            if (rc) return rc;
        }
        if (S.want_all)
            HIPCHK(hipMemcpyAsync(pout, A->dev + S.d_out, S.out_bytes, hipMemcpyDeviceToHost, A->stream));
        HIPCHK(hipMemcpyAsync(pw + W.out_begin, dw + W.out_begin, W.out_bytes, hipMemcpyDeviceToHost,
                              A->stream));
        HIPCHK(wait_stream(A->stream));
        if (intr_raised()) return fail(LDSR_EINTERRUPTED, "interrupted by the caller's interrupt callback");
        if (S.want_all) slice_scatter(S, pout);
        const double nan = std::numeric_limits<double>::quiet_NaN();
        memcpy(S.h_winner, pw + W.cell, sizeof(int) * ns);
        memcpy(S.h_theta_w, pw + W.theta, sizeof(double) * (size_t)ns * P);
        memcpy(S.h_lik_w, pw + W.lik_w, sizeof(double) * ns);
        memcpy(S.h_nit_w, pw + W.nit_w, sizeof(int) * ns);
        if (S.h_liks_w) memcpy(S.h_liks_w, pw + W.liks, sizeof(double) * (size_t)ns * S.niter);
        double *rows[4] = {S.h_X, S.h_Y, S.h_V, S.h_J};
        const size_t roff[4] = {W.X, W.Y, W.V, W.J};
        for (int k = 0; k < 4; k++) {
            if (!rows[k]) continue;
            memcpy(rows[k], pw + roff[k], sizeof(double) * (size_t)ns * T);
            for (int i = 0; i < ns; i++)      // series without a winner: the fit kernel skipped them
                if (S.h_winner[i] < 0)
                    for (int t = 0; t < T; t++) rows[k][(size_t)i * T + t] = nan;
        }
        S.fused_done = true;
        return LDSR_OK;
    }
    HIPCHK(hipMemcpyAsync(pout, A->dev + S.d_out, S.out_bytes, hipMemcpyDeviceToHost, A->stream));
    HIPCHK(wait_stream(A->stream));
    slice_scatter(S, pout);
    if (S.liks)     // the full trace goes straight to the caller's (pageable) array, series by series
        for (int s = 0; s < S.n_series; s++) {
            const size_t nc = (size_t)(S.off[(size_t)s + 1] - S.off[(size_t)s]);
            if (nc)
                HIPCHK(hipMemcpy(S.liks + (size_t)S.g_lo[(size_t)s] * S.niter,
                                 A->dev + S.d_liks + sizeof(double) * (size_t)S.off[(size_t)s] * S.niter,
                                 sizeof(double) * nc * S.niter, hipMemcpyDeviceToHost));
        }
    return LDSR_OK;
}

// Phase 2.  w_series / w_cell: local series index and local cell index of each winner of this
// slice; outputs are rows [i] of the given host arrays (any of liks_w .. J may be NULL).
static int slice_fit_winners(Slice &S, int n_w, const int *w_series, const int *w_cell,
                             double *liks_w, double *X, double *Y, double *V, double *J) {
    if (n_w == 0) return LDSR_OK;
    Arena *A = S.lease.a;
    const int T = S.T, P = S.P, niter = S.niter;
    HIPCHK(hipSetDevice(S.device));
    const WinLayout W = win_layout(std::max(S.max_winners, 1), P, T, niter);
    char *dw = A->dev + S.d_w, *pw = A->pin + S.p_w;
    memcpy(pw + W.cell, w_cell, sizeof(int) * n_w);
    memcpy(pw + W.ser, w_series, sizeof(int) * n_w);
    HIPCHK(hipMemcpyAsync(dw + W.cell, pw + W.cell, sizeof(int) * n_w, hipMemcpyHostToDevice, A->stream));
    HIPCHK(hipMemcpyAsync(dw + W.ser, pw + W.ser, sizeof(int) * n_w, hipMemcpyHostToDevice, A->stream));
    GatherParams gp;
    gp.n_w = n_w; gp.P = P; gp.niter = niter;
    gp.cell = (const int *)(dw + W.cell);
    gp.theta = (const double *)(A->dev + S.d_theta);
    gp.theta0 = (const double *)(A->dev + S.d_th0);
    gp.n_iter = (const int *)(A->dev + S.d_nit);
    gp.liks = S.trace_on_device ? (const double *)(A->dev + S.d_liks) : nullptr;
    gp.theta_w = (double *)(dw + W.theta);
    gp.theta0_w = (double *)(dw + W.theta0);
    gp.liks_w = (double *)(dw + W.liks);
    gp.lik = nullptr; gp.lik_w = nullptr; gp.n_iter_w = nullptr; gp.blk = nullptr;
    HIPCHK(launch_gather_winners(gp, A->stream));
    char *ws = A->dev + S.d_ws;
    if (!S.trace_on_device && liks_w) {
        // the per-cell traces were too large to keep: re-run the winners alone (one cell per
        // series; a cell's result does not depend on its position in the grid) with a trace
        std::vector<int> sel_off((size_t)S.n_series + 1, 0), order((size_t)n_w);
        for (int i = 0; i < n_w; i++) sel_off[(size_t)w_series[i] + 1] = 1;
        for (int s = 0; s < S.n_series; s++) sel_off[(size_t)s + 1] += sel_off[(size_t)s];
        for (int i = 1; i < n_w; i++)
            if (w_series[i] <= w_series[i - 1]) return fail(LDSR_EINVAL, "internal: winners must be sorted by series");
        const size_t nuv = S.shared_uv ? 1 : (size_t)S.n_series;
        (void)nuv;
        int rc = em_batch_device_impl(
            S.device, A->stream, S.n_series, T, S.p, S.q, (const double *)(A->dev + S.d_y),
            S.u ? (const double *)(A->dev + S.d_u) : nullptr, S.v ? (const double *)(A->dev + S.d_v) : nullptr,
            S.shared_uv, sel_off.data(), (const double *)(dw + W.theta0), niter, S.tol,
            S.algo_used ? S.algo_used : S.algo,       // the kernel of the batch run, whatever AUTO would pick for a few cells
            (double *)(A->dev + S.d_theta), (double *)(A->dev + S.d_lik), (int *)(A->dev + S.d_nit),
            (int *)(A->dev + S.d_st), (double *)(dw + W.liks), 1, ws, S.wsb, nullptr, S.dense_hint, nullptr, -1,
            nullptr, S.lead_used);
        if (rc) return rc;
    }
    // the winners' fit: one smoother pass at theta_w on the prepared series
    if (X || Y || V || J) {
        std::vector<int> soc(w_series, w_series + n_w);
        int rc = launch_smoother(S.device, A->stream, T, S.p, S.q, S.PP, S.QQ, S.u != nullptr,
                                 S.v != nullptr, S.shared_uv, ws, S.L, n_w, soc,
                                 (const double *)(dw + W.theta), 1, 0, 0.0, (double *)(dw + W.X),
                                 (double *)(dw + W.Y), (double *)(dw + W.V), (double *)(dw + W.J),
                                 (double *)(dw + W.lik), nullptr, (int *)(dw + W.st),
                                 (int *)(dw + W.ser), false);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(pw + W.out_begin, dw + W.out_begin, W.out_bytes, hipMemcpyDeviceToHost,
                          A->stream));
    HIPCHK(hipStreamSynchronize(A->stream));
    for (int i = 0; i < n_w; i++) {
        if (liks_w) memcpy(liks_w + (size_t)i * niter, pw + W.liks + sizeof(double) * (size_t)i * niter, sizeof(double) * niter);
        const size_t row = sizeof(double) * (size_t)i * T;
        if (X) memcpy(X + (size_t)i * T, pw + W.X + row, sizeof(double) * T);
        if (Y) memcpy(Y + (size_t)i * T, pw + W.Y + row, sizeof(double) * T);
        if (V) memcpy(V + (size_t)i * T, pw + W.V + row, sizeof(double) * T);
        if (J) memcpy(J + (size_t)i * T, pw + W.J + row, sizeof(double) * T);
    }
    return LDSR_OK;
}

// Cut the cell grid over the devices BY SERIES: device d gets the d-th of n_devices contiguous
// parts of EVERY series' restarts.  Series differ a lot in iterations to converge (config 5: 34 k
// to 130 k E-steps per series), so contiguous ranges of the flattened grid -- round 2's cut -- left
// the devices up to 1.32x apart at eight; the reference hands each restart to whichever worker is
// idle (R/LDS_reconstruction.R:46), and restarts of one series are statistically alike, so equal
// shares of every series are equal shares of the work (1.004 / 1.007 / 1.012 at 2 / 4 / 8 on the
// same grid; tests/test_shard_gloo.py pins the bound).  Every slice carries all series (<= 100 KB
// each) and the whole call's offsets for AUTO's plan.
static void make_slices(std::vector<Slice> &sl, int n_devices, const int *devices, int n_series,
                        int T, int p, int q, const double *y, const double *u, const double *v,
                        int shared_uv, const int *cell_offsets, const double *theta0, int niter,
                        double tol, int algo, double *theta, double *lik, int *n_iter,
                        int *status, double *liks) {
    sl.resize((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) {
        Slice &S = sl[(size_t)d];
        S.device = devices[d];
        S.T = T; S.p = p; S.q = q; S.shared_uv = shared_uv; S.niter = niter; S.tol = tol; S.algo = algo;
        S.off.assign((size_t)n_series + 1, 0);
        S.g_lo.assign((size_t)n_series, 0);
        for (int s = 0; s < n_series; s++) {
            const long long a = cell_offsets[s], ns = cell_offsets[s + 1] - cell_offsets[s];
            // part (d + s) mod n_devices of series s: the remainders of restart counts the device count
            // does not divide rotate over the devices (ldsr_amd/shard.py rank_slice is the same rule)
            const long long k = (d + s) % n_devices;
            const int lo = (int)(a + ns * k / n_devices), hi = (int)(a + ns * (k + 1) / n_devices);
            S.g_lo[(size_t)s] = lo;
            S.off[(size_t)s + 1] = S.off[(size_t)s] + (hi - lo);
        }
        S.n_series = S.off[(size_t)n_series] > 0 ? n_series : 0;
        S.y = y; S.u = u; S.v = v;
        S.theta0 = theta0;
        S.theta = theta; S.lik = lik; S.n_iter = n_iter; S.status = status; S.liks = liks;
        if (n_devices > 1) { S.plan_off = cell_offsets; S.plan_ns = n_series; }
    }
}

// run phase 1 of every slice, one host thread per slice beyond the first
static int run_slices(std::vector<Slice> &sl) {
    const size_t n = sl.size();
    std::vector<int> rcs(n, LDSR_OK);
    std::vector<std::string> msgs(n);
    std::atomic<int> running{0};
    auto work = [&](size_t d, bool worker) {
        if (worker) t_worker = true;
        rcs[d] = slice_run(sl[d]);
        if (rcs[d]) msgs[d] = g_err;      // thread-local message of this worker
        if (worker) running.fetch_sub(1);
    };
    std::vector<std::thread> pool;
    for (size_t d = 1; d < n; d++)
        if (sl[d].n_series > 0) {
            running.fetch_add(1);
            pool.emplace_back(work, d, true);
        }
    if (n > 0 && sl[0].n_series > 0) work(0, false);
    while (t_poll && running.load() > 0) {     // keep the interrupt callback alive while joining
        intr_poll();
        usleep(200);
    }
    for (auto &t : pool) t.join();
    if (intr_raised()) return fail(LDSR_EINTERRUPTED, "interrupted by the caller's interrupt callback");
    for (size_t d = 0; d < n; d++)
        if (rcs[d]) return fail(rcs[d], "device " + std::to_string(sl[d].device) + ": " + msgs[d]);
    return LDSR_OK;
}

extern "C" int ldsr_em_batch_multi(int n_devices, const int *devices, int n_series, int T, int p,
                                   int q, const double *y, const double *u, const double *v,
                                   int shared_uv, const int *cell_offsets, const double *theta0,
                                   int niter, double tol, int algo, double *theta, double *lik,
                                   int *n_iter, int *status, double *liks) {
    if (n_devices < 1 || !devices) return fail(LDSR_EINVAL, "n_devices must be >= 1");
    int rc = check_common(n_series, T, p, q, y, cell_offsets);
    if (rc) return rc;
    rc = check_em(niter, tol);
    if (rc) return rc;
    if (!theta0 || !theta || !lik || !n_iter || !status) return fail(LDSR_EINVAL, "NULL pointer");
    if (cell_offsets[n_series] == 0) return LDSR_OK;
    IntrScope intr;
    std::vector<Slice> sl;
    make_slices(sl, n_devices, devices, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, theta0,
                niter, tol, algo, theta, lik, n_iter, status, liks);
    return run_slices(sl);
}

extern "C" int ldsr_em_batch(int device, int n_series, int T, int p, int q, const double *y,
                             const double *u, const double *v, int shared_uv,
                             const int *cell_offsets, const double *theta0, int niter, double tol,
                             int algo, double *theta, double *lik, int *n_iter, int *status,
                             double *liks) {
    return ldsr_em_batch_multi(1, &device, n_series, T, p, q, y, u, v, shared_uv, cell_offsets,
                               theta0, niter, tol, algo, theta, lik, n_iter, status, liks);
}

extern "C" int ldsr_em_restart_grid(int n_devices, const int *devices, int n_series, int T, int p,
                                    int q, const double *y, const double *u, const double *v,
                                    int shared_uv, const int *cell_offsets, const double *theta0,
                                    int niter, double tol, int algo, double *theta_all,
                                    double *lik_all, int *n_iter_all, int *status_all, int *winner,
                                    double *theta_w, double *lik_w, int *n_iter_w, double *liks_w,
                                    double *X, double *Y, double *V, double *J) {
    if (n_devices < 1 || !devices) return fail(LDSR_EINVAL, "n_devices must be >= 1");
    int rc = check_common(n_series, T, p, q, y, cell_offsets);
    if (rc) return rc;
    rc = check_em(niter, tol);
    if (rc) return rc;
    if (!theta0 || !winner || !theta_w || !lik_w || !n_iter_w)
        return fail(LDSR_EINVAL, "theta0, winner, theta_w, lik_w and n_iter_w must not be NULL");
    const int n_cells = cell_offsets[n_series];
    const int P = 6 + p + q;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    IntrScope intr;
    if (n_devices > n_cells) n_devices = n_cells > 0 ? n_cells : 1;
    // One slice: selection and the winners' fit are fused behind the EM kernel on the device and
    // the per-cell arrays cross PCIe only if asked for.  Several slices: per-cell results come
    // back first (temporaries if the caller did not ask), the host selects, then phase 2 runs on
    // the slice that owns each winner.
    const bool want_all = theta_all || lik_all || n_iter_all || status_all;
    const bool fused = n_devices == 1 && n_cells > 0 &&
                       sizeof(double) * (size_t)n_cells * niter <= liks_trace_cap_bytes();
    std::vector<double> t_theta, t_lik;
    std::vector<int> t_nit, t_st;
    if (!fused || want_all) {
        if (!theta_all) { t_theta.resize((size_t)n_cells * P + 1); theta_all = t_theta.data(); }
        if (!lik_all) { t_lik.resize((size_t)n_cells + 1); lik_all = t_lik.data(); }
        if (!n_iter_all) { t_nit.resize((size_t)n_cells + 1); n_iter_all = t_nit.data(); }
        if (!status_all) { t_st.resize((size_t)n_cells + 1); status_all = t_st.data(); }
    }
    std::vector<Slice> sl;
    make_slices(sl, n_devices, devices, n_series, T, p, q, y, u, v, shared_uv, cell_offsets, theta0,
                niter, tol, algo, theta_all, lik_all, n_iter_all, status_all, nullptr);
    for (Slice &S : sl) S.max_winners = S.n_series;
    if (fused) {
        Slice &S = sl[0];
        S.fuse = true;
        S.want_all = want_all;
        S.h_winner = winner; S.h_theta_w = theta_w; S.h_lik_w = lik_w; S.h_nit_w = n_iter_w;
        S.h_liks_w = liks_w; S.h_X = X; S.h_Y = Y; S.h_V = V; S.h_J = J;
    }
    if (n_cells > 0) {
        rc = run_slices(sl);
        if (rc) return rc;
        if (fused) {
            if (!sl[0].fused_done) return fail(LDSR_EINVAL, "internal: fused restart path did not run");
            return LDSR_OK;     // winner[] is already global (one slice: lo = 0)
        }
    }
    // selection (R/LDS_reconstruction.R:50-58), per series over its restarts
    for (int s = 0; s < n_series; s++) {
        const int a = cell_offsets[s], b = cell_offsets[s + 1];
        const int k = ldsr_select_restart(b - a, lik_all + a, theta_all + (size_t)a * P, p, q);
        winner[s] = k < 0 ? -1 : a + k;
        if (k < 0) {
            for (int j = 0; j < P; j++) theta_w[(size_t)s * P + j] = nan;
            lik_w[s] = nan;
            n_iter_w[s] = 0;
        } else {
            memcpy(theta_w + (size_t)s * P, theta_all + (size_t)(a + k) * P, sizeof(double) * P);
            lik_w[s] = lik_all[a + k];
            n_iter_w[s] = n_iter_all[a + k];
        }
        if (k < 0) {
            if (liks_w) for (int i = 0; i < niter; i++) liks_w[(size_t)s * niter + i] = nan;
            double *rows[4] = {X, Y, V, J};
            for (double *r : rows)
                if (r) for (int t = 0; t < T; t++) r[(size_t)s * T + t] = nan;
        }
    }
    if (!liks_w && !X && !Y && !V && !J) return LDSR_OK;
    // phase 2 on the slice that owns each winner
    for (size_t d = 0; d < sl.size(); d++) {
        Slice &S = sl[d];
        if (S.n_series == 0) continue;
        std::vector<int> ws_, wc_, gs_;
        for (int s = 0; s < S.n_series; s++) {
            const int lo = S.g_lo[(size_t)s], nc = S.off[(size_t)s + 1] - S.off[(size_t)s];
            if (winner[s] >= lo && winner[s] < lo + nc) {
                ws_.push_back(s);
                wc_.push_back(S.off[(size_t)s] + winner[s] - lo);
                gs_.push_back(s);
            }
        }
        const int n_w = (int)ws_.size();
        if (!n_w) continue;
        std::vector<double> b_liks, b_X, b_Y, b_V, b_J;
        if (liks_w) b_liks.resize((size_t)n_w * niter);
        if (X) b_X.resize((size_t)n_w * T);
        if (Y) b_Y.resize((size_t)n_w * T);
        if (V) b_V.resize((size_t)n_w * T);
        if (J) b_J.resize((size_t)n_w * T);
        rc = slice_fit_winners(S, n_w, ws_.data(), wc_.data(), liks_w ? b_liks.data() : nullptr,
                               X ? b_X.data() : nullptr, Y ? b_Y.data() : nullptr,
                               V ? b_V.data() : nullptr, J ? b_J.data() : nullptr);
        if (rc) return rc;
        for (int i = 0; i < n_w; i++) {
            const size_t s = (size_t)gs_[(size_t)i];
            if (liks_w) memcpy(liks_w + s * niter, b_liks.data() + (size_t)i * niter, sizeof(double) * niter);
            if (X) memcpy(X + s * T, b_X.data() + (size_t)i * T, sizeof(double) * T);
            if (Y) memcpy(Y + s * T, b_Y.data() + (size_t)i * T, sizeof(double) * T);
            if (V) memcpy(V + s * T, b_V.data() + (size_t)i * T, sizeof(double) * T);
            if (J) memcpy(J + s * T, b_J.data() + (size_t)i * T, sizeof(double) * T);
        }
    }
    return LDSR_OK;
}

extern "C" int ldsr_em_restart_groups(int n_devices, const int *devices, int n_groups,
                                      ldsr_group *groups, int niter, double tol, int algo) {
    if (n_devices < 1 || !devices) return fail(LDSR_EINVAL, "n_devices must be >= 1");
    if (n_groups < 0 || (n_groups > 0 && !groups)) return fail(LDSR_EINVAL, "groups must not be NULL");
    std::vector<std::string> msgs((size_t)n_groups);
    IntrScope intr;
    std::atomic<int> running{0};
    auto work = [&](int g) {
        ldsr_group &G = groups[g];
        // rotate the device list so that concurrent groups start on different GPUs
        std::vector<int> devs((size_t)n_devices);
        for (int d = 0; d < n_devices; d++) devs[(size_t)d] = devices[(g + d) % n_devices];
        G.rc = ldsr_em_restart_grid(n_devices, devs.data(), G.n_series, G.T, G.p, G.q, G.y, G.u, G.v,
                                    G.shared_uv, G.cell_offsets, G.theta0, niter, tol, algo,
                                    G.theta_all, G.lik_all, G.n_iter_all, G.status_all, G.winner,
                                    G.theta_w, G.lik_w, G.n_iter_w, G.liks_w, G.X, G.Y, G.V, G.J);
        if (G.rc) msgs[(size_t)g] = g_err;
    };
    // a bounded pool: at most 8 groups in flight (each holds one arena per device)
    const int n_workers = std::min(n_groups, 8);
    std::vector<std::thread> pool;
    for (int w = 0; w < n_workers; w++) {
        running.fetch_add(1);
        pool.emplace_back([&, w]() {
            t_worker = true;
            for (int g = w; g < n_groups; g += n_workers) work(g);
            running.fetch_sub(1);
        });
    }
    while (t_poll && running.load() > 0) {     // the caller's thread keeps the interrupt callback alive
        intr_poll();
        usleep(200);
    }
    for (auto &t : pool) t.join();
    if (intr_raised()) return fail(LDSR_EINTERRUPTED, "interrupted by the caller's interrupt callback");
    for (int g = 0; g < n_groups; g++)
        if (groups[g].rc) return fail(groups[g].rc, "group " + std::to_string(g) + ": " + msgs[(size_t)g]);
    return LDSR_OK;
}

// ---- smoother / propagate / mstep / penalized likelihood host entry points ---------------------
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
    if (mode == 2 && (!X || !V || !J || !theta_out)) return fail(LDSR_EINVAL, "mstep needs X, V, J and theta");
    if (mode != 2 && (!theta_in || !lik)) return fail(LDSR_EINVAL, "theta and lik must not be NULL");
    const int P = 6 + p + q;
    const int PP = ldsr_pad_dim(p), QQ = ldsr_pad_dim(q);
    const size_t nuv = shared_uv ? 1 : (size_t)n_series;
    const size_t nT = (size_t)n_cells * T;
    const bool scalar_only = mode == 3;     // only [n_cells] scalars leave the device
    ArenaLease lease;
    rc = arena_acquire(device, &lease.a);
    if (rc) return rc;
    Arena *A = lease.a;
    WsLayout L = ws_layout(n_series, T, PP, QQ, shared_uv, n_cells, LDSR_ALGO_SCAN, 1);
    L.img2_stride = 0;       // no pair-family launch here: series_prep builds the scan image only
    L.img3_stride = 0;
    // the serial smoother uses X / V as its filtered-state strip; the scan FIT kernel needs none
    const bool need_strip = !(mode == 0 || mode == 3) || !em_scan_supported(T, PP, QQ);
    Carver c;
    const size_t o_y = c.take(sizeof(double) * (size_t)n_series * T);
    const size_t o_u = c.take(u ? sizeof(double) * nuv * T * p : 0);
    const size_t o_v = c.take(v ? sizeof(double) * nuv * T * q : 0);
    const size_t o_th = c.take(sizeof(double) * (size_t)n_cells * P);
    const size_t o_soc = c.take(sizeof(int) * (size_t)n_cells);
    const size_t in_bytes = c.o;
    const size_t o_lik = c.take(sizeof(double) * (size_t)n_cells);
    const size_t o_pen = c.take(sizeof(double) * (size_t)n_cells);
    const size_t o_st = c.take(sizeof(int) * (size_t)n_cells);
    const size_t o_tho = c.take(sizeof(double) * (size_t)n_cells * P);
    const size_t small_out = c.o - o_lik;
    const size_t o_X = c.take((scalar_only && !need_strip) ? 0 : sizeof(double) * nT);
    const size_t o_V = c.take((scalar_only && !need_strip) ? 0 : sizeof(double) * nT);
    const size_t o_Y = c.take(scalar_only ? 0 : sizeof(double) * nT);
    const size_t o_J = c.take(scalar_only ? 0 : sizeof(double) * nT);
    const size_t o_ws = c.take(L.total);
    rc = arena_reserve(A, c.o, in_bytes + align256(small_out));
    if (rc) return rc;
    char *dev = A->dev, *pin = A->pin;
    memcpy(pin + o_y, y, sizeof(double) * (size_t)n_series * T);
    if (u) memcpy(pin + o_u, u, sizeof(double) * nuv * T * p);
    if (v) memcpy(pin + o_v, v, sizeof(double) * nuv * T * q);
    if (theta_in) memcpy(pin + o_th, theta_in, sizeof(double) * (size_t)n_cells * P);
    int *soc = (int *)(pin + o_soc);
    for (int s = 0; s < n_series; s++)
        for (int cc = cell_offsets[s]; cc < cell_offsets[s + 1]; cc++) soc[cc] = s;
    HIPCHK(hipMemcpyAsync(dev, pin, in_bytes, hipMemcpyHostToDevice, A->stream));
    char *ws = dev + o_ws;
    rc = prepare_series(A->stream, n_series, T, p, q, PP, QQ, (const double *)(dev + o_y),
                        u ? (const double *)(dev + o_u) : nullptr,
                        v ? (const double *)(dev + o_v) : nullptr, shared_uv, ws, L);
    if (rc) return rc;

    std::vector<int> soc_v(soc, soc + n_cells);
    char *pout = pin + in_bytes;
    if (mode == 2) {
        // the fit arrives from the caller: X, V, J rows straight into the device arrays
        HIPCHK(hipMemcpyAsync(dev + o_X, X, sizeof(double) * nT, hipMemcpyHostToDevice, A->stream));
        HIPCHK(hipMemcpyAsync(dev + o_V, V, sizeof(double) * nT, hipMemcpyHostToDevice, A->stream));
        HIPCHK(hipMemcpyAsync(dev + o_J, J, sizeof(double) * nT, hipMemcpyHostToDevice, A->stream));
        SmoothParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.T = T; sp.p = p; sp.q = q; sp.has_u = u != nullptr; sp.has_v = v != nullptr;
        sp.n_cells = n_cells;
        sp.yp = (const double *)(ws + L.yp);
        sp.up = (const double *)(ws + L.up);
        sp.vp = (const double *)(ws + L.vp);
        sp.u_stride = shared_uv ? 0 : (long)T * PP;
        sp.v_stride = shared_uv ? 0 : (long)T * QQ;
        sp.sc = (const SeriesConst *)(ws + L.sc);
        sp.series_of_cell = (const int *)(dev + o_soc);
        sp.X = (double *)(dev + o_X); sp.V = (double *)(dev + o_V); sp.J = (double *)(dev + o_J);
        sp.status = (int *)(dev + o_st);
        sp.theta_out = (double *)(dev + o_tho);
        HIPCHK(launch_mstep(sp, PP, QQ, A->stream));
        HIPCHK(hipMemcpyAsync(pout, dev + o_lik, small_out, hipMemcpyDeviceToHost, A->stream));
        HIPCHK(hipStreamSynchronize(A->stream));
        memcpy(theta_out, pout + (o_tho - o_lik), sizeof(double) * (size_t)n_cells * P);
        if (status) memcpy(status, pout + (o_st - o_lik), sizeof(int) * (size_t)n_cells);
        return LDSR_OK;
    }
    rc = launch_smoother(device, A->stream, T, p, q, PP, QQ, u != nullptr, v != nullptr, shared_uv,
                         ws, L, n_cells, soc_v, (const double *)(dev + o_th), stdlik,
                         mode == 1 ? 1 : 0, lambda, (double *)(dev + o_X), (double *)(dev + o_Y),
                         (double *)(dev + o_V), (double *)(dev + o_J), (double *)(dev + o_lik),
                         scalar_only ? (double *)(dev + o_pen) : nullptr, (int *)(dev + o_st),
                         (int *)(dev + o_soc), scalar_only);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(pout, dev + o_lik, small_out, hipMemcpyDeviceToHost, A->stream));
    HIPCHK(hipStreamSynchronize(A->stream));
    if (scalar_only) {
        memcpy(lik, pout + (o_pen - o_lik), sizeof(double) * (size_t)n_cells);
        return LDSR_OK;
    }
    memcpy(lik, pout, sizeof(double) * (size_t)n_cells);
    if (X) HIPCHK(hipMemcpy(X, dev + o_X, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (Y) HIPCHK(hipMemcpy(Y, dev + o_Y, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (V) HIPCHK(hipMemcpy(V, dev + o_V, sizeof(double) * nT, hipMemcpyDeviceToHost));
    if (J && mode == 0) HIPCHK(hipMemcpy(J, dev + o_J, sizeof(double) * nT, hipMemcpyDeviceToHost));
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
