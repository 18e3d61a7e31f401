// ldsr_kernels.h -- host-visible launch interface between the C ABI (ldsr_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "ldsr_device.h"

struct PrepParams {
    int T, p, q, PP, QQ, shared_uv;
    const double *y, *u, *v;  // raw inputs as handed over the ABI (u / v may be null)
    double *yp, *up, *vp;     // prepared copies in the workspace
    double *yz;               // y with 0 where missing
    double *img;              // scan kernel: chunk-transposed series images [n_series][img_stride], or null
    long img_stride;          // doubles per image
    int L, NL;                // chunk length and virtual lanes (64 W) of the image layout
    double *img2;             // pair kernel: second image per series (NL2 virtual lanes, chunk length L2), or null
    long img2_stride;
    int L2, NL2;
    int lead;                 // pair kernel's LEAD form: img2 covers the tail [lead, T) only ...
    double *img3;             // ... and img3 the (whitened) u_t of the lead, [step of the lane][NL2 lanes][PP], or null
    long img3_stride;
    SeriesConst *sc;
    int *queue;               // [n_series] work-queue heads, reset to 0 here
    // Cell order of the pair kernel's steady form (em_pair_impl.h em_pair_body_steady): n_series more
    // workgroups sort every series' cells by the predicted number of steps its variance recursion needs
    // to settle at theta0, slowest first -- perm[position] = cell.  Null: no ordering.
    int *perm;                // [n_cells]
    int *perm_key;            // [n_cells] scratch (the cells' buckets)
    const int *cell_off;      // [n_series + 1] device copy of the cell offsets
    const double *theta0;     // [n_cells][6 + p + q]
    int n_series;
    int order_cpb;            // static schedule: cells per workgroup (positions are dealt across the workgroups); 0 = work queue
    int order_ntr;            // steps of the transient block (L - 1)
};

struct SmoothParams {
    int T, p, q, has_u, has_v, n_cells, stdlik, mode;
    const double *yp, *up, *vp;
    long u_stride, v_stride;
    const SeriesConst *sc;
    const int *series_of_cell;
    const double *theta;  // [n_cells][P] input (smooth / propagate)
    double *X, *Y, *V, *J, *lik;
    double *theta_out;    // mstep
    int *status;
    double *pen;          // smoother: lik - lambda * ssq (penalized_likelihood), may be null
    double lambda;
    int scalar_only;      // only lik / pen leave the kernel: Y and J are not written
};

// Restart selection on the device (R/LDS_reconstruction.R:50-58), one workgroup per series.
struct SelectParams {
    int n_series, P, c_index;     // c_index: position of C in a packed theta (1 + p)
    const int *off;               // [n_series + 1] cell offsets
    const double *theta, *lik;    // [n_cells][P], [n_cells]
    int *winner;                  // [n_series] cell index or -1
};

// Winner extraction of the restart-grid entry: row i of the *_w arrays = cell[i] of the batch
// (cell[i] < 0: no winner -- NaN rows).  Optionally emits the FIT launch's block table
// (series i, cell i, one cell or none) and the winners' lik / n_iter.
struct GatherParams {
    int n_w, P, niter;
    const int *cell;          // [n_w] cell index of each winner, or -1
    const double *theta;      // [n_cells][P] fitted
    const double *theta0;     // [n_cells][P] initial
    const int *n_iter;        // [n_cells]
    const double *liks;       // [n_cells][niter] traces (entries beyond n_iter undefined), or null
    double *theta_w, *theta0_w;   // [n_w][P]
    double *liks_w;           // [n_w][niter], NaN padded (untouched when liks is null)
    const double *lik;        // [n_cells] (with lik_w)
    double *lik_w;            // [n_w] or null
    int *n_iter_w;            // [n_w] or null
    int *blk;                 // [3][n_w] block table (series, first cell, cells) or null
};

static inline int ldsr_pad_dim(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 16; }

hipError_t launch_series_prep(const PrepParams &prm, int n_series, hipStream_t stream);
hipError_t launch_em_serial(const EmParams &prm, int PP, int QQ, int n_blocks, hipStream_t stream);
hipError_t launch_em_scan(const EmParams &prm, int PP, int QQ, int n_blocks, bool queue, bool fit,
                          hipStream_t stream);
void em_scan_layout(int T, int PP, int QQ, int *L, int *NL, long *img_doubles);
bool em_scan_global_image(int T, int PP, int QQ);   // series image too large for LDS: read from L2
bool em_scan_queue_only(int T, int PP, int QQ);     // shapes compiled with the work-queue schedule only
bool em_scan_supported(int T, int PP, int QQ);
int em_scan_cells_per_block(int T, int PP, int QQ);
// two (lpc = 32) or four (lpc = 16) cells per wave (em_pair_impl.h): T <= lpc * 32, padded p, q <= 4
bool em_pair_supported(int T, int PP, int QQ, int lpc, bool lead_form = false);   // lead_form: T = the tail of a closed-form lead
int em_pair_cells_per_block(int T, int PP, int QQ, int lpc, int lead = 0);   // T: the steps the sweeps work on
int em_pair_waves_per_block(int T, int PP, int QQ, int lpc, int lead = 0);
void em_pair_layout(int T, int PP, int QQ, int lpc, int *L, long *img_doubles, bool lead_form = false);
hipError_t launch_em_pair(const EmParams &prm, int PP, int QQ, int lpc, int n_blocks, bool queue, hipStream_t stream);
void em_pair_kernel_name(int T, int PP, int QQ, int lpc, bool queue, char *buf, size_t len, bool lead = false);
// kernel names as rocprofv3 prints them (ldsr_em_plan)
void em_scan_kernel_name(int T, int PP, int QQ, bool queue, bool fit, char *buf, size_t len);
void em_serial_kernel_name(int T, int PP, int QQ, char *buf, size_t len);
#include <string>
void em_kernel_inventory(std::string &out);      // names of every compiled scan / pair instantiation, one per line
hipError_t launch_gather_winners(const GatherParams &prm, hipStream_t stream);
hipError_t launch_select_winners(const SelectParams &prm, hipStream_t stream);
hipError_t launch_smooth(const SmoothParams &prm, int PP, int QQ, hipStream_t stream);
hipError_t launch_mstep(const SmoothParams &prm, int PP, int QQ, hipStream_t stream);
