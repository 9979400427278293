// P1 finite-element assembly on triangles for gfx950 (the step BEFORE the
// hot path).  Replaces the per-element Python loops of the reference
//   assembly/StiffnessMatrix.py:21-36 (compute_stiffness_2d), MassMatrix.py:21-35,
//   LoadVector.py:20-34
// which do `A[np.ix_(l2g, l2g)] += loc_A` element by element.
//
// Node-centric and atomic-free: thread i owns row i of A and M and entry i of rhs, walks the
// elements incident to node i IN ELEMENT ORDER (node->element adjacency built once from conn)
// and adds each element's contribution to its own row -- every matrix entry therefore
// receives its contributions in the same order as the reference's element loop, and two runs
// give identical bits.  The integrals use the reference's 3-point rule (Quadrature2D(3),
// Quadrature.py:84-98, including its 14-digit constants).
#include "lmg_common.hpp"

namespace {

__global__ void __launch_bounds__(256) p1_assemble_2d_kernel(
    int64_t n_nodes, const double *px, const double *py, const int *conn, const int *n2e_ptr,
    const int *n2e_elem, const int *n2e_loc, const double *coeff, double f_const, const int *rowptr,
    const int *colidx, double *a_vals, double *m_vals, double *rhs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const int rs = rowptr[i], re = rowptr[i + 1];
    for (int p = rs; p < re; ++p) {
        if (a_vals) a_vals[p] = 0.0;
        if (m_vals) m_vals[p] = 0.0;
    }
    // the reference's quadrature points (xi, eta) and weight
    const double qx[3] = {0.16666666666667, 0.16666666666667, 0.66666666666667};
    const double qy[3] = {0.16666666666667, 0.66666666666667, 0.16666666666667};
    const double w = 1.0 / 6.0;
    double r = 0.0;
    for (int t = n2e_ptr[i]; t < n2e_ptr[i + 1]; ++t) {
        const int e = n2e_elem[t], a = n2e_loc[t];
        const int v[3] = {conn[3 * e], conn[3 * e + 1], conn[3 * e + 2]};
        const double x0 = px[v[0]], x1 = px[v[1]], x2 = px[v[2]];
        const double y0 = py[v[0]], y1 = py[v[1]], y2 = py[v[2]];
        const double det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
        // gradients of the barycentric basis times det
        const double gx[3] = {y1 - y2, y2 - y0, y0 - y1};
        const double gy[3] = {x2 - x1, x0 - x2, x1 - x0};
        const double ke = coeff ? coeff[e] : 1.0;
        double phi_sum = 0.0;
        double mloc[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < 3; ++k) {
            const double ph[3] = {1.0 - qx[k] - qy[k], qx[k], qy[k]};
            phi_sum += w * ph[a];
            for (int bb = 0; bb < 3; ++bb) mloc[bb] += w * ph[a] * ph[bb];
        }
        r += det * (phi_sum * f_const);
        for (int bb = 0; bb < 3; ++bb) {
            const int col = v[bb];
            int p = rs;
            while (p < re && colidx[p] != col) ++p;
            if (p == re) continue;                      // pattern does not hold this entry
            if (a_vals) a_vals[p] += ke * ((gx[a] * gx[bb] + gy[a] * gy[bb]) / (2.0 * det));
            if (m_vals) m_vals[p] += det * mloc[bb];
        }
    }
    if (rhs) rhs[i] = r;
}

}  // namespace

extern "C" int lmg_p1_assemble_2d(int64_t n_nodes, const double *px, const double *py, const int32_t *conn,
                                  const int32_t *n2e_ptr, const int32_t *n2e_elem, const int32_t *n2e_loc,
                                  const double *coeff, double f_const, const int32_t *rowptr,
                                  const int32_t *colidx, double *a_vals, double *m_vals, double *rhs,
                                  void *stream)
{
    if (n_nodes < 0) return LMG_ERR_ARG;
    if (n_nodes == 0) return LMG_OK;
    if (!px || !py || !conn || !n2e_ptr || !n2e_elem || !n2e_loc || !rowptr || !colidx) return LMG_ERR_ARG;
    if (!a_vals && !m_vals && !rhs) return LMG_ERR_ARG;
    const unsigned grid = (unsigned)((n_nodes + 255) / 256);
    hipLaunchKernelGGL(p1_assemble_2d_kernel, dim3(grid), dim3(256), 0, lmg_stream(stream), n_nodes, px, py, conn,
                       n2e_ptr, n2e_elem, n2e_loc, coeff, f_const, rowptr, colidx, a_vals, m_vals, rhs);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}
