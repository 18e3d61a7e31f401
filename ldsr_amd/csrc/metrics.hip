// metrics.hip -- the five skill metrics of the reference's second translation unit, host code.
// They are not on the hot path (O(n) flops on the n ~ 12..46 points of a cross-validation fold),
// but a DLL that REPLACES ldsr.so has to keep them registered (/root/reference/src/RcppExports.cpp:
// 137-141), so the C ABI carries them and the .Call shim wraps them.
// Restated from the definitions in /root/reference/src/utils.cpp (NSE :13-22, nRMSE :36-39, corr
// :49-57, KGE :68-79, RE :93-97): the mean is Rcpp sugar's two-pass long-double mean, sd the
// (n - 1) standard deviation around it.  Pinned by the reference-held NPcv object through
// tests/test_npcv_fixture.py / tests/test_metrics_abi.py.
#include <cmath>

#include "../../include/ldsr_hip.h"

static double mean_of(int n, const double *x) {
    long double s = 0.0L;
    for (int i = 0; i < n; i++) s += x[i];
    s /= n;
    long double t = 0.0L;                      // second pass: the rounding left in the first
    for (int i = 0; i < n; i++) t += x[i] - s;
    return (double)(s + t / n);
}

static double sd_of(int n, const double *x) {
    const double m = mean_of(n, x);
    double ss = 0.0;
    for (int i = 0; i < n; i++) ss += (x[i] - m) * (x[i] - m);
    return std::sqrt(ss / (n - 1));
}

static double rss_of(int n, const double *yhat, const double *y) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += (y[i] - yhat[i]) * (y[i] - yhat[i]);
    return s;
}

extern "C" double ldsr_metric_nse(int n, const double *yhat, const double *y) {
    const double ybar = mean_of(n, y);
    double tss = 0.0;
    for (int i = 0; i < n; i++) tss += (y[i] - ybar) * (y[i] - ybar);
    return 1.0 - rss_of(n, yhat, y) / tss;
}

// (src/utils.cpp:37: sqrt(mean((y - yhat) * (y - yhat))) -- Rcpp sugar's mean over the squared
// residuals, i.e. the two-pass long-double mean, not sum / n)
extern "C" double ldsr_metric_nrmse(int n, const double *yhat, const double *y, double norm_const) {
    long double s = 0.0L;
    for (int i = 0; i < n; i++) s += (y[i] - yhat[i]) * (y[i] - yhat[i]);
    s /= n;
    long double t = 0.0L;
    for (int i = 0; i < n; i++) t += (y[i] - yhat[i]) * (y[i] - yhat[i]) - s;
    return std::sqrt((double)(s + t / n)) / norm_const;
}

extern "C" double ldsr_metric_corr(int n, const double *x, const double *y) {
    const double xbar = mean_of(n, x), ybar = mean_of(n, y), sx = sd_of(n, x), sy = sd_of(n, y);
    double s = 0.0;
    for (int i = 0; i < n; i++) s += ((x[i] - xbar) / sx) * ((y[i] - ybar) / sy);
    return s / (n - 1);
}

extern "C" double ldsr_metric_kge(int n, const double *yhat, const double *y) {
    const double r = ldsr_metric_corr(n, yhat, y);
    const double alpha = sd_of(n, yhat) / sd_of(n, y);
    const double beta = mean_of(n, yhat) / mean_of(n, y);
    return 1.0 - std::sqrt((r - 1.0) * (r - 1.0) + (alpha - 1.0) * (alpha - 1.0) + (beta - 1.0) * (beta - 1.0));
}

extern "C" double ldsr_metric_re(int n, const double *yhat, const double *y, double yc_bar) {
    double tss = 0.0;
    for (int i = 0; i < n; i++) tss += (y[i] - yc_bar) * (y[i] - yc_bar);
    return 1.0 - rss_of(n, yhat, y) / tss;
}
