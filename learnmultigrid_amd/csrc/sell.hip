// Sliced-ELL sweeps for gfx950: residual / Jacobi / SpMV on matrices with LONG rows.
//
// The Galerkin operators of L2-type / learned transfer operators have 25-50 entries per row
// with all-distinct values.  pcsr.hip stages a tile's entry stream through LDS and lets lane t
// walk row t; with rows that long a tile of 64 rows already needs ~30 KB of LDS (5 waves per
// CU) and -- worse -- lane t reads LDS at a stride of one ROW (48 entries = 96 dwords: all 64
// lanes land on two banks), so those levels ran at 1.8-3.0 TB/s.  Here the order of the
// entries in memory is changed instead (SELL-64, a lossless re-encoding like PCSR):
//   slice s = rows 64 s .. 64 s + 63, padded to the longest row L_s of the slice;
//   entry j of row r is stored at  slice_base[s] + 64 j + (r mod 64)
// so the wave that owns a slice reads entry j of all its 64 rows with ONE coalesced load per
// stream (columns: uint16 relative to the slice's smallest column when every slice spans
// < 65536 columns, else int32; values: raw fp64), straight into registers: no LDS, no barrier,
// full occupancy.  Lane t still accumulates row t in storage order with separately rounded
// products => bit-identical to the CSR kernels.  Padding costs (L_s - len) entries per row:
// a few per cent for the near-uniform rows this format is used for.
#include "lmg_common.hpp"
#include <limits.h>
#include <string.h>

namespace {

enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2 };
constexpr int kBlock = 256;
constexpr int kSlicesPerBlock = kBlock / LMG_WAVE;

struct SArgs {
    int n;
    int nslices;
    int nblocks;
    int blocks_per_xcd;
    const long long *slice_base;   // nslices: first padded entry of the slice
    const int *slice_len;          // nslices: padded row length L_s
    const int *slice_cmin;         // nslices: column base (COL16)
    const int *rowlen;             // n
    const void *col;
    const double *val;
    const double *x;
    const double *b;
    double *out;
    double alpha, beta;
    double *partial;
};

// NT: the entry streams (columns, values) are read once per sweep: nontemporal loads keep them from evicting the
// vector lines the gathers of x live on (L2)
template <int MODE, bool COL16, int JU, bool NT = false>
__global__ void __launch_bounds__(kBlock) sell_sweep_kernel(SArgs a)
{
    __shared__ double s_red[kBlock / LMG_WAVE];
    typedef typename std::conditional<COL16, unsigned short, int>::type col_t;
    const col_t *col = static_cast<const col_t *>(a.col);
    const int lane = threadIdx.x & (LMG_WAVE - 1), wave = threadIdx.x / LMG_WAVE;
    // XCD-aware: XCD k (blockIdx mod 8) works on the k-th contiguous eighth of the slices
    const int blk = (int)(blockIdx.x & 7u) * a.blocks_per_xcd + (int)(blockIdx.x >> 3);
    const int slice = blk * kSlicesPerBlock + wave;
    double local = 0.0;
    if (blk < a.nblocks && slice < a.nslices) {
        const int row = slice * LMG_WAVE + lane;
        const bool valid = row < a.n;
        const int len = valid ? a.rowlen[row] : 0;
        const int slen = a.slice_len[slice];
        const long long base = a.slice_base[slice] + lane;
        const int cb = COL16 ? a.slice_cmin[slice] : 0;
        const double bv = (MODE != MODE_SPMV && valid) ? a.b[row] : 0.0;
        double acc = 0.0, diag = 0.0, xi = 0.0;
        for (int j0 = 0; j0 < slen; j0 += JU) {
            int c[JU];
            double v[JU], xv[JU];
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
                const int j = (j0 + jj < slen) ? j0 + jj : slen - 1;      // padded storage: always readable
                const long long e = base + (long long)j * LMG_WAVE;
                c[jj] = cb + (int)(NT ? __builtin_nontemporal_load(col + e) : col[e]);
                v[jj] = NT ? __builtin_nontemporal_load(a.val + e) : a.val[e];
            }
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) xv[jj] = a.x[(j0 + jj < len) ? c[jj] : 0];
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
                const bool act = j0 + jj < len;
                const double s2 = acc + v[jj] * xv[jj];
                acc = act ? s2 : acc;
                if (MODE == MODE_JACOBI) {
                    const bool dg = act && c[jj] == row;
                    diag = dg ? diag + v[jj] : diag;
                    xi = dg ? xv[jj] : xi;
                }
            }
        }
        if (valid) {
            if (MODE == MODE_RESIDUAL) {
                const double r = bv - acc;
                if (a.out) a.out[row] = r;
                local = r * r;
            } else if (MODE == MODE_JACOBI) {
                const double r = bv - acc;
                if (diag != 0.0) a.out[row] = xi + a.alpha * ((1.0 / diag) * r);
                else a.out[row] = a.x[row];
            } else {
                double s = acc;
                if (a.alpha != 1.0) s = a.alpha * s;
                if (a.beta == 0.0) a.out[row] = s;
                else if (a.beta == 1.0) a.out[row] = a.out[row] + s;
                else a.out[row] = a.beta * a.out[row] + s;
            }
        }
    }
    if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
        const double tot = lmg_block_sum<kBlock>(local, s_red);
        if (threadIdx.x == 0 && blk < a.nblocks) a.partial[blk] = tot;
    }
}

__global__ void __launch_bounds__(1024) sell_reduce_partials_kernel(const double *partial, int64_t count, double *out)
{
    __shared__ double s_red[1024 / LMG_WAVE];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < count; i += 4 * 1024) {
        v0 += partial[i];
        v1 += partial[i + 1024];
        v2 += partial[i + 2048];
        v3 += partial[i + 3072];
    }
    for (; i < count; i += 1024) v0 += partial[i];
    const double tot = lmg_block_sum<1024>((v0 + v1) + (v2 + v3), s_red);
    if (threadIdx.x == 0) out[0] = tot;
}

// ---- building the format (setup) -------------------------------------------------------------
// one wave per slice: padded length and column range
__global__ void __launch_bounds__(kBlock) sell_slice_info_kernel(int64_t n, int64_t nslices, const int *__restrict__ rowptr,
                                                                 const int *__restrict__ colidx, int *slice_len,
                                                                 int *cmin, int *cmax)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t slice = (int64_t)blockIdx.x * kSlicesPerBlock + threadIdx.x / LMG_WAVE;
    if (slice >= nslices) return;
    const int64_t row = slice * LMG_WAVE + lane;
    int len = 0, mn = INT_MAX, mx = 0;
    if (row < n) {
        const int s = rowptr[row], e = rowptr[row + 1];
        len = e - s;
        for (int k = s; k < e; ++k) {
            const int c = colidx[k];
            mn = c < mn ? c : mn;
            mx = c > mx ? c : mx;
        }
    }
#pragma unroll
    for (int off = LMG_WAVE / 2; off > 0; off >>= 1) {
        const int l2 = __shfl_down(len, off, LMG_WAVE), a = __shfl_down(mn, off, LMG_WAVE), b = __shfl_down(mx, off, LMG_WAVE);
        len = l2 > len ? l2 : len;
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane == 0) {
        slice_len[slice] = len;
        cmin[slice] = mn == INT_MAX ? 0 : mn;
        cmax[slice] = mx;
    }
}

// lane = row: copies its entries to their padded positions (col_out may be null: values only)
template <bool COL16>
__global__ void __launch_bounds__(kBlock) sell_fill_kernel(int64_t n, int64_t nslices, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, const double *__restrict__ vals,
                                                           const long long *__restrict__ slice_base,
                                                           const int *__restrict__ cmin, void *col_out, double *val_out)
{
    typedef typename std::conditional<COL16, unsigned short, int>::type col_t;
    col_t *co = static_cast<col_t *>(col_out);
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t slice = (int64_t)blockIdx.x * kSlicesPerBlock + threadIdx.x / LMG_WAVE;
    if (slice >= nslices) return;
    const int64_t row = slice * LMG_WAVE + lane;
    if (row >= n) return;
    const int s = rowptr[row], e = rowptr[row + 1];
    const long long base = slice_base[slice] + lane;
    const int cb = COL16 ? cmin[slice] : 0;
    for (int k = s; k < e; ++k) {
        const long long p = base + (long long)(k - s) * LMG_WAVE;
        if (co) co[p] = (col_t)(colidx[k] - cb);
        val_out[p] = vals[k];
    }
}

// measured (tools/time_sell.py, Jacobi sweep): 2049^2 x 25 entries (1.05 GB) 0.264 -> 0.250 ms, 1025^2 x 49 (0.51 GB)
// 0.122 -> 0.114 ms, but 721^2 x 25 (0.13 GB: the whole matrix stays in the Infinity Cache) 0.025 -> 0.032 ms
int g_sell_ju = 0;                        // entries of a row in flight: 0 = by the longest row (5, 7 or 8); 5, 7, 8, 10 forced
int g_sell_nt = -1;                       // -1: by size, 0: never, 1: always
int64_t g_sell_nt_entries = 30000000;     // padded entries from which the streams no longer fit the 256 MB Infinity Cache

template <int MODE, bool COL16>
int launch(SArgs a, int max_len, hipStream_t st)
{
    // measured on MI355X (tools/tune_sweep.py --matrix L1|L2): see DESIGN.md
    const dim3 grid((unsigned)(a.blocks_per_xcd * 8)), block(kBlock);
    const bool nt = g_sell_nt == 1 || (g_sell_nt < 0 && (int64_t)a.n * max_len >= g_sell_nt_entries);
    // entries of a row in flight: the factor that leaves the fewest idle slots in the last trip over the longest row (25-entry
    // rows: 5 trips of 5 instead of 4 of 8 -- 2049^2 x 25: 0.2505 -> 0.2290 ms; 49: 7 x 7 -- 1025^2 x 49: 0.1057 -> 0.0982 ms)
    int ju = g_sell_ju;
    if (ju == 0) {
        ju = 8;
        int waste = (8 - max_len % 8) % 8;
        const int cand[2] = {7, 5};
        for (int k = 0; k < 2; ++k) {
            const int w = (cand[k] - max_len % cand[k]) % cand[k];
            if (w < waste) {
                waste = w;
                ju = cand[k];
            }
        }
    }
    if (max_len <= 12) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 4>), grid, block, 0, st, a);
    else if (ju == 5 && nt) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 5, true>), grid, block, 0, st, a);
    else if (ju == 5) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 5>), grid, block, 0, st, a);
    else if (ju == 7 && nt) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 7, true>), grid, block, 0, st, a);
    else if (ju == 7) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 7>), grid, block, 0, st, a);
    else if (ju == 10 && nt) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 10, true>), grid, block, 0, st, a);
    else if (ju == 10) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 10>), grid, block, 0, st, a);
    else if (nt) hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 8, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sell_sweep_kernel<MODE, COL16, 8>), grid, block, 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // namespace

int lmg_sell_tune_set(const char *key, int v)
{
    if (strcmp(key, "sell_nt") == 0) {
        if (v < -1 || v > 1) return LMG_ERR_ARG;
        g_sell_nt = v;
        return LMG_OK;
    }
    if (strcmp(key, "sell_ju") == 0) {
        if (v != 0 && v != 5 && v != 7 && v != 8 && v != 10) return LMG_ERR_ARG;
        g_sell_ju = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_sell_tune_get(const char *key)
{
    if (strcmp(key, "sell_nt") == 0) return g_sell_nt;
    if (strcmp(key, "sell_ju") == 0) return g_sell_ju;
    return LMG_ERR_ARG;
}

extern "C" {

int lmg_sell_sweep(int mode, int64_t n, const int64_t *slice_base, const int32_t *slice_len, const int32_t *slice_cmin,
                   const int32_t *rowlen, const void *col, int colmode, const double *val, int32_t max_len,
                   const double *x, const double *b, double *out, double alpha, double beta, double *partials,
                   double *norm2, void *stream)
{
    if (n < 0 || n >= INT32_MAX - 64 || (colmode != 0 && colmode != 1) || max_len < 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!slice_base || !slice_len || !rowlen || !col || !val || !x) return LMG_ERR_ARG;
    if (colmode == 0 && !slice_cmin) return LMG_ERR_ARG;
    if (mode == MODE_SPMV) {
        if (!out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_JACOBI) {
        if (!b || !out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_RESIDUAL) {
        if (!b || (partials == nullptr) != (norm2 == nullptr) || (!out && !partials)) return LMG_ERR_ARG;
    } else {
        return LMG_ERR_ARG;
    }
    SArgs a;
    a.n = (int)n;
    a.nslices = (int)((n + LMG_WAVE - 1) / LMG_WAVE);
    a.nblocks = (a.nslices + kSlicesPerBlock - 1) / kSlicesPerBlock;
    a.blocks_per_xcd = (a.nblocks + 7) / 8;
    a.slice_base = reinterpret_cast<const long long *>(slice_base);
    a.slice_len = slice_len;
    a.slice_cmin = slice_cmin;
    a.rowlen = rowlen;
    a.col = col;
    a.val = val;
    a.x = x;
    a.b = b;
    a.out = out;
    a.alpha = alpha;
    a.beta = beta;
    a.partial = (mode == MODE_RESIDUAL) ? partials : nullptr;
    hipStream_t st = lmg_stream(stream);
    int rc;
    if (colmode == 0) {
        rc = mode == MODE_RESIDUAL ? launch<MODE_RESIDUAL, true>(a, max_len, st)
           : mode == MODE_JACOBI ? launch<MODE_JACOBI, true>(a, max_len, st) : launch<MODE_SPMV, true>(a, max_len, st);
    } else {
        rc = mode == MODE_RESIDUAL ? launch<MODE_RESIDUAL, false>(a, max_len, st)
           : mode == MODE_JACOBI ? launch<MODE_JACOBI, false>(a, max_len, st) : launch<MODE_SPMV, false>(a, max_len, st);
    }
    if (rc != LMG_OK) return rc;
    if (mode == MODE_RESIDUAL && partials) {
        // blocks beyond nblocks (grid rounded up to a multiple of 8) write nothing: only nblocks partials exist
        hipLaunchKernelGGL(sell_reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partials, (int64_t)a.nblocks, norm2);
        LMG_CHECK_LAUNCH();
    }
    return LMG_OK;
}

int lmg_sell_slice_info(int64_t n, const int32_t *rowptr, const int32_t *colidx, int32_t *slice_len, int32_t *cmin,
                        int32_t *cmax, void *stream)
{
    if (n < 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!rowptr || !slice_len || !cmin || !cmax) return LMG_ERR_ARG;
    const int64_t nslices = (n + LMG_WAVE - 1) / LMG_WAVE;
    const unsigned grid = (unsigned)((nslices + kSlicesPerBlock - 1) / kSlicesPerBlock);
    hipLaunchKernelGGL(sell_slice_info_kernel, dim3(grid), dim3(kBlock), 0, lmg_stream(stream), n, nslices, rowptr,
                       colidx, slice_len, cmin, cmax);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_sell_fill(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                  const int64_t *slice_base, const int32_t *slice_cmin, int colmode, void *col_out, double *val_out,
                  void *stream)
{
    if (n < 0 || (colmode != 0 && colmode != 1)) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!rowptr || !vals || !slice_base || !val_out || (col_out && !colidx) || (colmode == 0 && !slice_cmin))
        return LMG_ERR_ARG;
    const int64_t nslices = (n + LMG_WAVE - 1) / LMG_WAVE;
    const unsigned grid = (unsigned)((nslices + kSlicesPerBlock - 1) / kSlicesPerBlock);
    const long long *sb = reinterpret_cast<const long long *>(slice_base);
    if (colmode == 0)
        hipLaunchKernelGGL(sell_fill_kernel<true>, dim3(grid), dim3(kBlock), 0, lmg_stream(stream), n, nslices, rowptr,
                           colidx, vals, sb, slice_cmin, col_out, val_out);
    else
        hipLaunchKernelGGL(sell_fill_kernel<false>, dim3(grid), dim3(kBlock), 0, lmg_stream(stream), n, nslices, rowptr,
                           colidx, vals, sb, slice_cmin, col_out, val_out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
