// Gauss-Seidel on gfx950.
//
// The reference's shipped smoother is pyamg's forward lexicographic sweep
// (Multigrid.py:88,:121): inherently sequential.  It is reproduced EXACTLY by level
// scheduling: rows are grouped into sets that are mutually independent in the pattern
// of A + A^T (dependencies AND anti-dependencies), sets are executed in order, and each
// row does the pyamg update in storage order.  The same set-executor run on colour
// classes gives multicolour Gauss-Seidel (the throughput variant; a different ordering
// whose CPU twin for the parity tests walks the same row list sequentially).
//
// Two executors:
//   * per-set launches (wide sets, e.g. the 4097-row anti-diagonals of a 4097^2 grid);
//   * one persistent workgroup that walks all sets with a workgroup barrier between
//     them (narrow sets: 1-D problems have n sets of ONE row; launching n kernels per
//     sweep would be pure launch latency).
#include <stdlib.h>
#include <vector>
#include "lmg_common.hpp"

namespace {

__device__ __forceinline__ void gs_update_row(int i, const int *rowptr, const int *colidx,
                                              const double *vals, double *x, const double *b)
{
    const int s = rowptr[i], e = rowptr[i + 1];
    double rsum = 0.0, diag = 0.0;
    for (int jj = s; jj < e; ++jj) {
        const int j = colidx[jj];
        const double v = vals[jj];
        if (j == i) diag = v;
        else rsum += v * x[j];
    }
    if (diag != 0.0) x[i] = (b[i] - rsum) / diag;
}

__global__ void __launch_bounds__(256) gs_rows_kernel(const int *rowptr, const int *colidx,
                                                      const double *vals, double *x, const double *b,
                                                      const int *rows, int64_t nrows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nrows) gs_update_row(rows[k], rowptr, colidx, vals, x, b);
}

// One workgroup, all sets, all sweeps.  Global stores of set s must be visible to the
// loads of set s+1 issued by other waves of the SAME workgroup (same CU): a
// workgroup-scope release/acquire fence around the barrier is sufficient for that.
constexpr int kSingleBlock = 1024;
int g_gs_single_max = 2 * kSingleBlock;     // widest set the one-workgroup executor takes (tunable)
__global__ void __launch_bounds__(kSingleBlock) gs_single_wg_kernel(
    const int *rowptr, const int *colidx, const double *vals, double *x, const double *b,
    const int *set_rows, const int *set_ptr, int64_t nsets, int sweeps)
{
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int64_t s = 0; s < nsets; ++s) {
            const int lo = set_ptr[s], hi = set_ptr[s + 1];
            for (int k = lo + (int)threadIdx.x; k < hi; k += kSingleBlock)
                gs_update_row(set_rows[k], rowptr, colidx, vals, x, b);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

// The same executor on a copy of the PATTERN in schedule order (ELL, column-major: entry j of
// scheduled row k at ell_cols[j * total + k]).  gs_single_wg_kernel pays four dependent memory
// round trips per set (row id -> row pointer -> column -> x); here the row id, its entry range
// and all its columns are one level of coalesced loads at a position every lane knows in
// advance, so that level for the first row of the NEXT set is issued before the barrier and
// only values + x gathers remain behind it.  Values are still read from the CSR array (they
// may change between sweeps: Galerkin rebuilds keep the schedule).  Same update, same order.
// X_IN_LDS: the whole iterate lives in LDS for the duration of the kernel (small grids, up to
// 18 000 unknowns): between two sets there is then only a workgroup barrier, no round trip
// through L2.
template <int K, bool X_IN_LDS>
__global__ void __launch_bounds__(kSingleBlock) gs_ell_single_wg_kernel(
    const double *vals, double *xg, const double *b, const int *ell_row, const int *ell_start,
    const int *ell_len, const int *ell_cols, int64_t total, const int *set_ptr, int64_t nsets, int sweeps, int n)
{
    extern __shared__ double s_xl[];
    const int tid = threadIdx.x;
    double *x = X_IN_LDS ? s_xl : xg;
    if (X_IN_LDS) {
        for (int i = tid; i < n; i += kSingleBlock) s_xl[i] = xg[i];
        __syncthreads();
    }
    int p_row = 0, p_st = 0, p_len = 0, p_c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) p_c[j] = 0;
    auto level1 = [&](int64_t k) {
        p_row = ell_row[k];
        p_st = ell_start[k];
        p_len = ell_len[k];
#pragma unroll
        for (int j = 0; j < K; ++j) p_c[j] = ell_cols[(int64_t)j * total + k];
    };
    auto relax = [&]() {
        double v[K], xv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            v[j] = vals[j < p_len ? p_st + j : 0];
            xv[j] = x[p_c[j]];                      // padding columns point at the row itself
        }
        const double bi = b[p_row];
        double rsum = 0.0, diag = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < p_len) {
                if (p_c[j] == p_row) diag = v[j];
                else rsum += v[j] * xv[j];
            }
        }
        if (diag != 0.0) x[p_row] = (bi - rsum) / diag;
    };
    {
        const int lo = set_ptr[0], hi = set_ptr[1];
        if (lo + tid < hi) level1(lo + tid);
    }
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int64_t s = 0; s < nsets; ++s) {
            const int lo = set_ptr[s], hi = set_ptr[s + 1];
            int k = lo + tid;
            if (k < hi) relax();                    // first row of the lane: prefetched before the barrier
            for (k += kSingleBlock; k < hi; k += kSingleBlock) {
                level1(k);
                relax();
            }
            // first row of the next set (wrapping to the next sweep), independent of x
            const int64_t ns = (s + 1 < nsets) ? s + 1 : 0;
            if (s + 1 < nsets || sw + 1 < sweeps) {
                const int nlo = set_ptr[ns], nhi = set_ptr[ns + 1];
                if (nlo + tid < nhi) level1(nlo + tid);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (X_IN_LDS) {
        for (int i = tid; i < n; i += kSingleBlock) xg[i] = s_xl[i];
    }
}

// Chain-like schedules (sets of one or two rows: 1-D problems, where the level schedule
// degenerates to the sequential sweep).  One wave; the row data of 64 consecutive rows is fetched by the
// 64 lanes at once (it does not depend on x), then the 64 rows are relaxed one after the
// other, lane l at step l, with x held in LDS when it fits (the only serial dependence is the
// x[i-1] -> x[i] hand-over: one LDS round trip instead of a workgroup barrier plus a global
// round trip per row).  Same row arithmetic and order as gs_update_row.
constexpr int kChainK = 8;     // entries of a row kept in registers; longer rows read the rest from memory
template <bool X_IN_LDS>
__global__ void __launch_bounds__(64) gs_chain_kernel(const int *rowptr, const int *colidx,
                                                      const double *vals, double *x, const double *b,
                                                      const int *rows, int nrows, int n, int sweeps)
{
    extern __shared__ double s_x[];
    const int lane = threadIdx.x;
    if (X_IN_LDS) {
        for (int i = lane; i < n; i += 64) s_x[i] = x[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int c0 = 0; c0 < nrows; c0 += 64) {
            const int k = c0 + lane;
            const bool valid = k < nrows;
            const int row = valid ? rows[k] : 0;
            const int rs = valid ? rowptr[row] : 0;
            const int len = valid ? rowptr[row + 1] - rs : 0;
            const double bi = valid ? b[row] : 0.0;
            int cj[kChainK];
            double vj[kChainK];
#pragma unroll
            for (int q = 0; q < kChainK; ++q) {
                cj[q] = (q < len) ? colidx[rs + q] : -1;
                vj[q] = (q < len) ? vals[rs + q] : 0.0;
            }
            const int steps = min(64, nrows - c0);
            for (int l = 0; l < steps; ++l) {
                if (lane == l) {
                    double rsum = 0.0, diag = 0.0;
#pragma unroll
                    for (int q = 0; q < kChainK; ++q) {
                        if (q < len) {
                            const int j = cj[q];
                            if (j == row) diag = vj[q];
                            else rsum += vj[q] * (X_IN_LDS ? s_x[j] : x[j]);
                        }
                    }
                    for (int q = kChainK; q < len; ++q) {
                        const int j = colidx[rs + q];
                        const double v = vals[rs + q];
                        if (j == row) diag = v;
                        else rsum += v * (X_IN_LDS ? s_x[j] : x[j]);
                    }
                    if (diag != 0.0) {
                        const double xn = (bi - rsum) / diag;
                        if (X_IN_LDS) s_x[row] = xn;
                        else x[row] = xn;
                    }
                }
                // hand-over to the next lane: same wave, but the compiler must not move the
                // next lane's reads above this lane's write
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
    }
    if (X_IN_LDS) {
        for (int i = lane; i < n; i += 64) x[i] = s_x[i];
    }
}

// Pattern of A^T (CSC of A) on the host: counting sort by column.
void transpose_pattern(int64_t n, const int32_t *Ap, const int32_t *Aj, std::vector<int32_t> &Tp,
                       std::vector<int32_t> &Tj)
{
    const int64_t nnz = Ap[n];
    Tp.assign(n + 1, 0);
    Tj.resize(nnz);
    for (int64_t k = 0; k < nnz; ++k)
        if (Aj[k] >= 0 && Aj[k] < n) Tp[Aj[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) Tp[i + 1] += Tp[i];
    std::vector<int32_t> next(Tp.begin(), Tp.end() - 1);
    for (int64_t i = 0; i < n; ++i)
        for (int32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
            const int32_t j = Aj[k];
            if (j >= 0 && j < n) Tj[next[j]++] = (int32_t)i;
        }
}

}  // namespace

int lmg_gs_tune_set(int v)
{
    if (v < 1) return LMG_ERR_ARG;
    g_gs_single_max = v;
    return LMG_OK;
}
int lmg_gs_tune_get(void) { return g_gs_single_max; }

extern "C" {

int lmg_csr_gs_rows(const int32_t *rp, const int32_t *ci, const double *va, double *x, const double *b,
                    const int32_t *rows, int64_t nrows, void *stream)
{
    if (nrows < 0 || !rp || !ci || !va || !x || !b || (nrows > 0 && !rows)) return LMG_ERR_ARG;
    if (nrows == 0) return LMG_OK;
    const int block = nrows >= 256 ? 256 : 64;
    const unsigned grid = (unsigned)((nrows + block - 1) / block);
    hipLaunchKernelGGL(gs_rows_kernel, dim3(grid), dim3(block), 0, lmg_stream(stream), rp, ci, va, x, b,
                       rows, nrows);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_csr_gs_schedule(const int32_t *rp, const int32_t *ci, const double *va, double *x,
                        const double *b, const int32_t *d_set_rows, const int32_t *d_set_ptr,
                        const int32_t *h_set_ptr, int64_t nsets, int64_t max_set, int sweeps,
                        void *stream)
{
    if (nsets < 0 || sweeps < 0 || !rp || !ci || !va || !x || !b) return LMG_ERR_ARG;
    if (nsets == 0 || sweeps == 0) return LMG_OK;
    if (!d_set_rows || !d_set_ptr || !h_set_ptr) return LMG_ERR_ARG;
    hipStream_t st = lmg_stream(stream);
    const int64_t total_rows = h_set_ptr[nsets];
    if (total_rows <= 4 * nsets && total_rows < (1 << 30)) {
        // (nearly) a chain: relaxing the rows one after the other in schedule order is the
        // same sweep (rows of one set are independent) and avoids a barrier per set
        const int n = (int)total_rows;
        const size_t lds = (size_t)n * sizeof(double);
        if (lds <= 144 * 1024) {
            // (set on every call: the attribute is per device, and a process may drive several)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gs_chain_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            hipLaunchKernelGGL(gs_chain_kernel<true>, dim3(1), dim3(64), lds, st, rp, ci, va, x, b, d_set_rows, n,
                               n, sweeps);
        } else {
            hipLaunchKernelGGL(gs_chain_kernel<false>, dim3(1), dim3(64), 0, st, rp, ci, va, x, b, d_set_rows, n,
                               n, sweeps);
        }
        LMG_CHECK_LAUNCH();
        return LMG_OK;
    }
    if (max_set <= g_gs_single_max) {
        hipLaunchKernelGGL(gs_single_wg_kernel, dim3(1), dim3(kSingleBlock), 0, st, rp, ci, va, x, b,
                           d_set_rows, d_set_ptr, nsets, sweeps);
        LMG_CHECK_LAUNCH();
        return LMG_OK;
    }
    for (int sw = 0; sw < sweeps; ++sw)
        for (int64_t s = 0; s < nsets; ++s) {
            const int64_t cnt = (int64_t)h_set_ptr[s + 1] - h_set_ptr[s];
            if (cnt <= 0) continue;
            const int block = cnt >= 256 ? 256 : 64;
            const unsigned grid = (unsigned)((cnt + block - 1) / block);
            hipLaunchKernelGGL(gs_rows_kernel, dim3(grid), dim3(block), 0, st, rp, ci, va, x, b,
                               d_set_rows + h_set_ptr[s], cnt);
        }
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_csr_gs_schedule_ell(int64_t n, const double *va, double *x, const double *b, const int32_t *d_ell_row,
                            const int32_t *d_ell_start, const int32_t *d_ell_len, const int32_t *d_ell_cols,
                            int32_t ell_k, int64_t total_rows, const int32_t *d_set_ptr, int64_t nsets, int sweeps,
                            void *stream)
{
    if (n < 0 || n >= INT32_MAX) return LMG_ERR_ARG;
    if (nsets < 0 || sweeps < 0 || total_rows < 0 || !va || !x || !b) return LMG_ERR_ARG;
    if (nsets == 0 || sweeps == 0 || total_rows == 0) return LMG_OK;
    if (!d_ell_row || !d_ell_start || !d_ell_len || !d_ell_cols || !d_set_ptr) return LMG_ERR_ARG;
    hipStream_t st = lmg_stream(stream);
    const size_t lds = (size_t)n * sizeof(double);
    const bool in_lds = lds <= 144 * 1024;
#define LMG_GS_ELL(KK)                                                                                               \
    do {                                                                                                             \
        if (in_lds) {                                                                                                \
            if (lds > 48 * 1024)                                                                                     \
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gs_ell_single_wg_kernel<KK, true>),         \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
            hipLaunchKernelGGL((gs_ell_single_wg_kernel<KK, true>), dim3(1), dim3(kSingleBlock), lds, st, va, x, b,  \
                               d_ell_row, d_ell_start, d_ell_len, d_ell_cols, total_rows, d_set_ptr, nsets, sweeps,  \
                               (int)n);                                                                              \
        } else {                                                                                                     \
            hipLaunchKernelGGL((gs_ell_single_wg_kernel<KK, false>), dim3(1), dim3(kSingleBlock), 0, st, va, x, b,   \
                               d_ell_row, d_ell_start, d_ell_len, d_ell_cols, total_rows, d_set_ptr, nsets, sweeps,  \
                               (int)n);                                                                              \
        }                                                                                                            \
    } while (0)
    switch (ell_k) {
    case 3: LMG_GS_ELL(3); break;
    case 5: LMG_GS_ELL(5); break;
    case 7: LMG_GS_ELL(7); break;
    case 9: LMG_GS_ELL(9); break;
    case 16: LMG_GS_ELL(16); break;
    default: return LMG_ERR_ARG;
    }
#undef LMG_GS_ELL
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int64_t lmg_host_gs_levels(int64_t n, const int32_t *Ap, const int32_t *Aj, int32_t *level)
{
    if (n < 0 || !Ap || (n > 0 && !level)) return LMG_ERR_ARG;
    if (n == 0) return 0;
    if (Ap[n] > 0 && !Aj) return LMG_ERR_ARG;
    std::vector<int32_t> Tp, Tj;
    transpose_pattern(n, Ap, Aj, Tp, Tj);
    int32_t nlev = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t lv = -1;
        for (int32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
            const int32_t j = Aj[k];
            if (j >= 0 && j < i && level[j] > lv) lv = level[j];
        }
        for (int32_t k = Tp[i]; k < Tp[i + 1]; ++k) {
            const int32_t j = Tj[k];
            if (j < i && level[j] > lv) lv = level[j];
        }
        level[i] = lv + 1;
        if (lv + 2 > nlev) nlev = lv + 2;
    }
    return nlev;
}

int64_t lmg_host_greedy_colors(int64_t n, const int32_t *Ap, const int32_t *Aj, int32_t *color)
{
    if (n < 0 || !Ap || (n > 0 && !color)) return LMG_ERR_ARG;
    if (n == 0) return 0;
    if (Ap[n] > 0 && !Aj) return LMG_ERR_ARG;
    std::vector<int32_t> Tp, Tj;
    transpose_pattern(n, Ap, Aj, Tp, Tj);
    std::vector<int64_t> mark;   // mark[c] == i  <=> colour c is taken by an earlier neighbour of i
    int32_t ncol = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int32_t k = Ap[i]; k < Ap[i + 1]; ++k) {
            const int32_t j = Aj[k];
            if (j >= 0 && j < i) mark[color[j]] = i;
        }
        for (int32_t k = Tp[i]; k < Tp[i + 1]; ++k) {
            const int32_t j = Tj[k];
            if (j < i) mark[color[j]] = i;
        }
        int32_t c = 0;
        while (c < ncol && mark[c] == i) ++c;
        if (c == ncol) {
            mark.push_back(-1);
            ++ncol;
        }
        color[i] = c;
    }
    return ncol;
}

}  // extern "C"
