// 1-D L2-projection coupling operator on the device (the step before the hot path).
//
// learn_multigrid/L2_projection: Intersection.find_intersections1d (an O(ne * ne_c) double loop,
// Intersection.py:60-76) + CouplingOperator.compute_b_1d (3-point Gauss quadrature on every intersection
// segment, CouplingOperator.py:32-69) + the row normalisations of L2Projection.py:74-90.  Here one thread owns
// one FINE NODE: the segments of its two elements are the pieces the coarse nodes cut out of them (two binary
// searches per element -- the nodes of both meshes are sorted), the row of B gets the contributions of those
// segments in left-to-right order (the order in which the reference adds them), and the "pseudo" (lumped
// mass) or "quasi" (row sum) scaling is applied in the same pass.  Rows are written as CSR: a counting pass,
// an exclusive scan by the caller, a filling pass.
#include "lmg_common.hpp"

namespace {

constexpr int kB = 256;
// Quadrature(3) of the reference (assembly/Quadrature.py:52-66), its 14-digit constants included
__device__ const double kGP[3] = {0.11270166537926, 0.50000000000000, 0.88729833462074};
__device__ const double kGW[3] = {0.27777777777778, 0.44444444444444, 0.27777777777778};

__device__ __forceinline__ int upper_bound(const double *x, int n, double v)      // first index with x[i] > v
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int lower_bound(const double *x, int n, double v)      // first index with x[i] >= v
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// fine element e = [xf[e], xf[e+1]]: the coarse nodes strictly inside it are xc[lo .. hi-1]; its first
// segment lies in coarse element ce0 (clamped like the host restatement does for nodes outside the coarse mesh)
struct Elem {
    int lo, hi, ce0;
};
__device__ __forceinline__ Elem elem_info(const double *xf, const double *xc, int nc, int e)
{
    Elem r;
    const double a = xf[e], b = xf[e + 1];
    r.lo = upper_bound(xc, nc, a);
    r.hi = lower_bound(xc, nc, b);
    if (r.hi < r.lo) r.hi = r.lo;
    r.ce0 = min(max(r.lo - 1, 0), nc - 2);
    return r;
}

// columns touched by the row of fine node i: [c_first, c_last]
__device__ __forceinline__ void row_range(const double *xf, const double *xc, int nf, int nc, int i, int &c_first, int &c_last)
{
    c_first = nc;
    c_last = -1;
    if (i > 0) {
        const Elem L = elem_info(xf, xc, nc, i - 1);
        c_first = min(c_first, L.ce0);
        c_last = max(c_last, min(L.ce0 + (L.hi - L.lo), nc - 2) + 1);
    }
    if (i < nf - 1) {
        const Elem R = elem_info(xf, xc, nc, i);
        c_first = min(c_first, R.ce0);
        c_last = max(c_last, min(R.ce0 + (R.hi - R.lo), nc - 2) + 1);
    }
}

__global__ void __launch_bounds__(kB) l2_count_kernel(int nf, int nc, const double *xf, const double *xc, int *rownnz)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nf) return;
    int c0, c1;
    row_range(xf, xc, nf, nc, i, c0, c1);
    rownnz[i] = c1 >= c0 ? c1 - c0 + 1 : 0;
}

// adds the contributions of fine element e (local node `ln` of it is the row's node) to out[col - c_first]
__device__ __forceinline__ void add_element(const double *xf, const double *xc, int nc, int e, int ln, int c_first,
                                            double *out)
{
    const Elem E = elem_info(xf, xc, nc, e);
    const double fa = xf[e], fb = xf[e + 1];
    const int nseg = E.hi - E.lo + 1;
    for (int s = 0; s < nseg; ++s) {
        const double xa = s == 0 ? fa : xc[E.lo + s - 1];
        const double xb = s == nseg - 1 ? fb : xc[E.lo + s];
        const int ce = min(E.ce0 + s, nc - 2);
        const double ca = xc[ce], cb = xc[ce + 1];
        double l0 = 0.0, l1 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double p = xa + kGP[k] * (xb - xa);                 // g_function
            const double fr = (p - fa) / (fb - fa);                   // inv_g_function
            const double cr = (p - ca) / (cb - ca);
            const double pf = ln == 0 ? 1.0 - fr : fr;
            l0 += pf * (1.0 - cr) * kGW[k];
            l1 += pf * cr * kGW[k];
        }
        out[ce - c_first] += (xb - xa) * l0;
        out[ce + 1 - c_first] += (xb - xa) * l1;
    }
}

// kind 0: B, 1: "pseudo" (B / lumped fine mass), 2: "quasi" (B / row sum)
__global__ void __launch_bounds__(kB) l2_fill_kernel(int kind, int nf, int nc, const double *xf, const double *xc,
                                                     const int *rowptr, int *colidx, double *vals)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nf) return;
    int c0, c1;
    row_range(xf, xc, nf, nc, i, c0, c1);
    const int s = rowptr[i], len = rowptr[i + 1] - s;
    if (len <= 0) return;
    double *out = vals + s;
    for (int j = 0; j < len; ++j) {
        out[j] = 0.0;
        colidx[s + j] = c0 + j;
    }
    if (i > 0) add_element(xf, xc, nc, i - 1, 1, c0, out);
    if (i < nf - 1) add_element(xf, xc, nc, i, 0, c0, out);
    double scale = 1.0;
    if (kind == 1) {
        // column sum of the P1 mass matrix (MassMatrix.compute_mass_1d with the same quadrature):
        // h_left * (m01 + m11) + h_right * (m00 + m10)
        double m00 = 0.0, m01 = 0.0, m11 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m00 += (1.0 - kGP[k]) * (1.0 - kGP[k]) * kGW[k];
            m01 += (1.0 - kGP[k]) * kGP[k] * kGW[k];
            m11 += kGP[k] * kGP[k] * kGW[k];
        }
        double lumped = 0.0;
        if (i > 0) lumped += (xf[i] - xf[i - 1]) * m01 + (xf[i] - xf[i - 1]) * m11;
        if (i < nf - 1) lumped += (xf[i + 1] - xf[i]) * m00 + (xf[i + 1] - xf[i]) * m01;
        scale = 1.0 / lumped;
        for (int j = 0; j < len; ++j) out[j] = out[j] / lumped;
        return;
    }
    if (kind == 2) {
        double rs = 0.0;
        for (int j = 0; j < len; ++j) rs += out[j];
        for (int j = 0; j < len; ++j) out[j] = out[j] / rs;
    }
    (void)scale;
}

}  // namespace

extern "C" {

int lmg_l2_coupling_count(int64_t nf, int64_t nc, const double *xf, const double *xc, int32_t *rownnz, void *stream)
{
    if (nf < 2 || nc < 2 || nf >= INT32_MAX || nc >= INT32_MAX || !xf || !xc || !rownnz) return LMG_ERR_ARG;
    l2_count_kernel<<<(unsigned)((nf + kB - 1) / kB), kB, 0, lmg_stream(stream)>>>((int)nf, (int)nc, xf, xc, rownnz);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_l2_coupling_fill(int kind, int64_t nf, int64_t nc, const double *xf, const double *xc, const int32_t *rowptr,
                         int32_t *colidx, double *vals, void *stream)
{
    if (kind < 0 || kind > 2 || nf < 2 || nc < 2 || nf >= INT32_MAX || nc >= INT32_MAX) return LMG_ERR_ARG;
    if (!xf || !xc || !rowptr || !colidx || !vals) return LMG_ERR_ARG;
    l2_fill_kernel<<<(unsigned)((nf + kB - 1) / kB), kB, 0, lmg_stream(stream)>>>(kind, (int)nf, (int)nc, xf, xc, rowptr,
                                                                                  colidx, vals);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
