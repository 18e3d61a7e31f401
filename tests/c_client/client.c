/* Plain-C client of include/ldsr_hip.h (no HIP / torch / Python): reads a case file
 *   T p q n niter tol
 *   y[T]  u[T*p] (time-major)  v[T*q]  theta0[n*(6+p+q)]
 * runs ldsr_em_batch over every visible GPU (ldsr_em_batch_multi), applies the reference's
 * selection rule and prints  winner n_iter lik  and the winner's theta.  "nan" in y = NA. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ldsr_hip.h"

static double *readv(FILE *f, size_t n) {
    double *a = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; i++)
        if (fscanf(f, "%lf", &a[i]) != 1) { fprintf(stderr, "short input\n"); exit(2); }
    return a;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "r");
    if (!f) return 2;
    int T, p, q, n, niter;
    double tol;
    if (fscanf(f, "%d %d %d %d %d %lf", &T, &p, &q, &n, &niter, &tol) != 6) return 2;
    const int P = 6 + p + q;
    double *y = readv(f, T), *u = readv(f, (size_t)T * p), *v = readv(f, (size_t)T * q);
    double *th0 = readv(f, (size_t)n * P);
    fclose(f);
    double *th = (double *)malloc(sizeof(double) * n * P), *lik = (double *)malloc(sizeof(double) * n);
    int *nit = (int *)malloc(sizeof(int) * n), *st = (int *)malloc(sizeof(int) * n);
    int ndev = ldsr_device_count();
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 3; }
    int *devs = (int *)malloc(sizeof(int) * ndev);
    for (int d = 0; d < ndev; d++) devs[d] = d;
    const int off[2] = {0, n};
    int rc = ldsr_em_batch_multi(ndev, devs, 1, T, p, q, y, u, v, 0, off, th0, niter, tol,
                                 LDSR_ALGO_AUTO, th, lik, nit, st, NULL);
    if (rc != LDSR_OK) { fprintf(stderr, "ldsr error %d: %s\n", rc, ldsr_last_error()); return 1; }
    const int k = ldsr_select_restart(n, lik, th, p, q);
    printf("%d %d %.17g\n", k, k >= 0 ? nit[k] : -1, k >= 0 ? lik[k] : NAN);
    for (int i = 0; k >= 0 && i < P; i++) printf("%.17g%c", th[(size_t)k * P + i], i + 1 < P ? ' ' : '\n');
    ldsr_shutdown();
    return 0;
}
