/*
 * oracle/lmg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the arithmetic on the reference's
 * V-cycle hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; learnmultigrid_amd/ never does.
 *
 * Built with -ffp-contract=off so every a*b+c is two roundings, exactly like
 * the SciPy / pyamg C++ loops the reference calls (those are compiled for
 * generic x86-64 without FMA contraction).  The HIP kernels are built with the
 * same flag and accumulate rows in the same order, which is what makes the
 * SpMV / Jacobi / Gauss-Seidel parity tests bit-exact rather than "close".
 *
 * Reference call sites restated here (paths relative to /root/reference):
 *   orc_csr_matvec     learn_multigrid/solvers/Multigrid.py:62,:90 (A.dot(u));
 *                      Jacobi.py:28; GaussSeidel.py:29  -- SciPy csr_matvec:
 *                      sum = y[i]; for jj in row: sum += Ax[jj]*x[Aj[jj]].
 *   orc_csr_residual   same lines: rhs - A.dot(u) (matvec first, then subtract).
 *   orc_csr_jacobi     learn_multigrid/solvers/Jacobi.py:22-35
 *                      (solution += inv_d * residual_vector, inv_d = 1/diag).
 *   orc_csr_gs_forward pyamg.relaxation.relaxation.gauss_seidel(sweep='forward')
 *                      as called at Multigrid.py:88,:121 (pyamg is NOT vendored in
 *                      the reference and not installed; this is a restatement of
 *                      its published amg_core::gauss_seidel loop -- "parity
 *                      unpinned" at that boundary, cross-checked against the
 *                      reference's own GaussSeidel.py:22-37 in tests).
 *   orc_csr_gs_rows    same row update, over an explicit row list (the CPU twin
 *                      of the multicolour / level-scheduled device orderings).
 */
#include <stddef.h>
#include <stdint.h>

void orc_csr_matvec(int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                    const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; ++jj)
            sum += Ax[jj] * x[Aj[jj]];
        y[i] = sum;
    }
}

/* y = alpha * (A x) + beta * y ; beta == 0 never reads y. */
void orc_csr_spmv(int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                  const double *x, double *y, double alpha, double beta)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; ++jj)
            sum += Ax[jj] * x[Aj[jj]];
        if (alpha != 1.0) sum = alpha * sum;
        if (beta == 0.0)      y[i] = sum;
        else if (beta == 1.0) y[i] = y[i] + sum;
        else                  y[i] = beta * y[i] + sum;
    }
}

/* r = b - A x ; returns sum_i r_i^2 accumulated left to right. */
double orc_csr_residual(int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                        const double *x, const double *b, double *r)
{
    double nrm2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; ++jj)
            sum += Ax[jj] * x[Aj[jj]];
        double ri = b[i] - sum;
        r[i] = ri;
        nrm2 += ri * ri;
    }
    return nrm2;
}

/* x_out = x + omega * ((1/a_ii) * (b - A x)); rows with a zero/missing diagonal keep x. */
void orc_csr_jacobi(int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                    const double *x, const double *b, double omega, double *x_out)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0, diag = 0.0;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            int32_t j = Aj[jj];
            if (j == (int32_t)i) diag += Ax[jj];
            sum += Ax[jj] * x[j];
        }
        double ri = b[i] - sum;
        if (diag != 0.0) {
            double upd = (1.0 / diag) * ri;
            if (omega != 1.0) upd = omega * upd;
            x_out[i] = x[i] + upd;
        } else {
            x_out[i] = x[i];
        }
    }
}

static inline void gs_row(int64_t i, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                          double *x, const double *b)
{
    double rsum = 0.0, diag = 0.0;
    for (int32_t jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
        int32_t j = Aj[jj];
        if (j == (int32_t)i) diag = Ax[jj];
        else                 rsum += Ax[jj] * x[j];
    }
    if (diag != 0.0) x[i] = (b[i] - rsum) / diag;
}

void orc_csr_gs_forward(int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                        double *x, const double *b, int32_t iterations)
{
    for (int32_t it = 0; it < iterations; ++it)
        for (int64_t i = 0; i < n; ++i)
            gs_row(i, Ap, Aj, Ax, x, b);
}

void orc_csr_gs_rows(const int32_t *Ap, const int32_t *Aj, const double *Ax,
                     double *x, const double *b, const int32_t *rows, int64_t nrows)
{
    for (int64_t k = 0; k < nrows; ++k)
        gs_row(rows[k], Ap, Aj, Ax, x, b);
}

/* Dense y = M x (row-major), the restatement used to check the device coarse solve. */
void orc_dense_gemv(int64_t n, int64_t m, const double *M, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int64_t j = 0; j < m; ++j) sum += M[i * m + j] * x[j];
        y[i] = sum;
    }
}
