// Vector kernels, halo pack/unpack, dense coarse-operator GEMV, scan, hipGraph helpers,
// and the small management entry points of the C ABI.
#include <string.h>
#include "lmg_common.hpp"

int lmg_sweep_tune_set(int rpt);
int lmg_sweep_tune_get(void);
int lmg_pcsr_tune_set(int ju);
int lmg_pcsr_tune_get(void);
int lmg_rpat_tune_set(int v);
int lmg_rpat_tune_get(void);
int lmg_rpat_nt_set(int v);
int lmg_rpat_nt_get(void);
int lmg_stencil_tune_set(const char *key, int v);
int lmg_stencil_tune_get(const char *key);
int lmg_fused_tune_set(const char *key, int v);
int lmg_fused_tune_get(const char *key);
int lmg_tile_tune_set(const char *key, int v);
int lmg_tile_tune_get(const char *key);
int lmg_dia_tune_set(const char *key, int v);
int lmg_dia_tune_get(const char *key);
int lmg_sell_tune_set(const char *key, int v);
int lmg_sell_tune_get(const char *key);
int lmg_gsw_tune_set(const char *key, int v);
int lmg_gsw_tune_get(const char *key);
int lmg_gs_tune_set(int v);
int lmg_gs_tune_get(void);

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 8;   // 256 CUs x 8 resident workgroups: grid-stride beyond

inline unsigned grid_for(int64_t n, int per_block)
{
    int64_t g = (n + per_block - 1) / per_block;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// y = alpha*x + beta*y, two elements (16 bytes) per lane per step.
__global__ void __launch_bounds__(kBlock) axpby_kernel(int64_t n, double alpha, const double *x,
                                                       double beta, double *y)
{
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    double2 *y2 = reinterpret_cast<double2 *>(y);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        const double2 xv = x2[i];
        double2 yv;
        if (beta == 0.0) {
            yv.x = alpha * xv.x;
            yv.y = alpha * xv.y;
        } else {
            yv = y2[i];
            yv.x = alpha * xv.x + beta * yv.x;
            yv.y = alpha * xv.y + beta * yv.y;
        }
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        y[i] = (beta == 0.0) ? alpha * x[i] : alpha * x[i] + beta * y[i];
    }
}

// out = alpha * (x * y) elementwise: the first Jacobi sweep from a zero initial guess,
// x_1 = omega * (D^-1 b), bit-identical to the general sweep with x_0 = 0.
__global__ void __launch_bounds__(kBlock) vmul_kernel(int64_t n, double alpha, const double *x,
                                                      const double *y, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = alpha * (x[i] * y[i]);
}

// fixed geometry (1024 partials) so the result does not depend on tuning knobs
__global__ void __launch_bounds__(kBlock) dot_kernel(int64_t n, const double *x, const double *y,
                                                     double *partial)
{
    __shared__ double s_red[kBlock / LMG_WAVE];
    double v = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) v += x[i] * y[i];
    const double tot = lmg_block_sum<kBlock>(v, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) dot_final_kernel(const double *partial, int count, double *out)
{
    __shared__ double s_red[1024 / LMG_WAVE];
    double v = (int)threadIdx.x < count ? partial[threadIdx.x] : 0.0;
    const double tot = lmg_block_sum<1024>(v, s_red);
    if (threadIdx.x == 0) out[0] = tot;
}

__global__ void __launch_bounds__(kBlock) gather_kernel(int64_t n, const int *idx, const double *x,
                                                        double *buf)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) buf[i] = x[idx[i]];
}

__global__ void __launch_bounds__(kBlock) scatter_kernel(int64_t n, const int *idx, const double *buf,
                                                         double *x)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[idx[i]] = buf[i];
}

// Dense y = M x, one wave per row, 16 bytes per lane per step, fixed summation order.
// HBM-bound: 8*n*m bytes of M per call.
// With bs > 0 the matrix is block diagonal, stored as n/bs dense bs x m blocks one after the
// other: row r multiplies the x segment of its block, x[(r/bs)*m .. +m).
__global__ void __launch_bounds__(kBlock) dense_gemv_kernel(int64_t n, int64_t m, const double *M,
                                                            const double *x0, double *y, int64_t bs)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LMG_WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * kBlock / LMG_WAVE;
    const bool vec_ok = (m % 2 == 0);
    for (int64_t row = wave; row < n; row += nwaves) {
        const double *Mr = M + row * m;
        const double *x = bs > 0 ? x0 + (row / bs) * m : x0;
        double s = 0.0;
        if (vec_ok) {
            const double2 *M2 = reinterpret_cast<const double2 *>(Mr);
            const double2 *x2 = reinterpret_cast<const double2 *>(x);
            for (int64_t j = lane; j < m / 2; j += LMG_WAVE) {
                const double2 mv = M2[j], xv = x2[j];
                s += mv.x * xv.x;
                s += mv.y * xv.y;
            }
        } else {
            for (int64_t j = lane; j < m; j += LMG_WAVE) s += Mr[j] * x[j];
        }
        s = lmg_wave_sum(s);
        if (lane == 0) y[row] = s;
    }
}

// The same product with one WORKGROUP (4 waves) per row: a few thousand long rows (the Schur-complement
// inverse of the banded coarse solver, 2967 x 2967) give one wave per row too little to keep 256 CUs busy
// (26 us for 70 MB; 4 waves per row: see DESIGN.md).  The four partial sums are added in wave order, so the
// result does not depend on the launch geometry of other rows -- but it differs in rounding from the
// one-wave kernel, which is why the choice between the two only depends on (n, m).
__global__ void __launch_bounds__(kBlock) dense_gemv_wgrow_kernel(int64_t n, int64_t m, const double *M,
                                                                  const double *x, double *y)
{
    __shared__ double s_part[kBlock / LMG_WAVE];
    const int t = threadIdx.x, lane = t & (LMG_WAVE - 1), w = t / LMG_WAVE;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const double2 *M2 = reinterpret_cast<const double2 *>(M + row * m);
        const double2 *x2 = reinterpret_cast<const double2 *>(x);
        double s = 0.0;
        // all loads of a chunk are issued before its sums (the plain loop waits for every load in turn: a row of a
        // few thousand entries is 5 steps per lane, i.e. 5 memory latencies); same order of additions
        constexpr int CH = 8;
        for (int64_t j0 = t; j0 < m / 2; j0 += (int64_t)CH * kBlock) {
            double2 mv[CH], xv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int64_t j = j0 + (int64_t)c * kBlock;
                const bool in = j < m / 2;
                mv[c] = M2[in ? j : 0];
                xv[c] = x2[in ? j : 0];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (j0 + (int64_t)c * kBlock < m / 2) {
                    s += mv[c].x * xv[c].x;
                    s += mv[c].y * xv[c].y;
                }
            }
        }
        if ((m & 1) && t == 0) s += M[row * m + m - 1] * x[m - 1];
        s = lmg_wave_sum(s);
        if (lane == 0) s_part[w] = s;
        __syncthreads();
        if (t == 0) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < kBlock / LMG_WAVE; ++q) tot += s_part[q];
            y[row] = tot;
        }
        __syncthreads();
    }
}

// Batched GEMV on windows of a vector: for block k and row r
//     y[k*ys + r] = (z ? z[k*zs + r] : 0) + alpha * sum_c M[k][r][c] * x[k*xs + c]
// (M: nb dense rows x cols blocks, row-major, one after the other).  One wave per row, like
// dense_gemv_kernel; the windows of neighbouring blocks may overlap (xs < cols).  This is the only
// kernel of the block-cyclic-reduction coarse solver (coarse.py BlockCyclicReduction).
__global__ void __launch_bounds__(kBlock) dense_gemv_windows_kernel(int64_t nb, int64_t rows, int64_t cols,
                                                                    const double *M, const double *x0, int64_t xs,
                                                                    const double *z0, int64_t zs, double alpha,
                                                                    double *y0, int64_t ys)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LMG_WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * kBlock / LMG_WAVE;
    for (int64_t row = wave; row < nb * rows; row += nwaves) {
        const int64_t k = row / rows, r = row - k * rows;
        const double2 *M2 = reinterpret_cast<const double2 *>(M + row * cols);
        const double2 *x2 = reinterpret_cast<const double2 *>(x0 + k * xs);
        double s = 0.0;
        for (int64_t j = lane; j < cols / 2; j += LMG_WAVE) {
            const double2 mv = M2[j], xv = x2[j];
            s += mv.x * xv.x;
            s += mv.y * xv.y;
        }
        s = lmg_wave_sum(s);
        if (lane == 0) y0[k * ys + r] = (z0 ? z0[k * zs + r] : 0.0) + alpha * s;
    }
}

// The same with a table of window starts instead of a stride: x window of block k = x0[xoff[k] .. xoff[k] + cols)
// (the banded coarse solver's back-substitution x_I = y_I - (A_II^-1 A_IS) x_S: strip k couples to the separators on
// either side of it, which start at irregular offsets; 8-byte aligned windows are enough for the 16-byte loads).
__global__ void __launch_bounds__(kBlock) dense_gemv_windows_off_kernel(int64_t nb, int64_t rows, int64_t cols,
                                                                        const double *M, const double *x0, const int *xoff,
                                                                        const double *z0, int64_t zs, double alpha,
                                                                        double *y0, int64_t ys)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LMG_WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * kBlock / LMG_WAVE;
    for (int64_t row = wave; row < nb * rows; row += nwaves) {
        const int64_t k = row / rows, r = row - k * rows;
        const double2 *M2 = reinterpret_cast<const double2 *>(M + row * cols);
        const double *x = x0 + xoff[k];
        double s = 0.0;
        for (int64_t j = lane; j < cols / 2; j += LMG_WAVE) {
            const double2 mv = M2[j];
            s += mv.x * x[2 * j];
            s += mv.y * x[2 * j + 1];
        }
        s = lmg_wave_sum(s);
        if (lane == 0) y0[k * ys + r] = (z0 ? z0[k * zs + r] : 0.0) + alpha * s;
    }
}

// First and last product of the banded coarse solver with the permutation [strips | separators] folded in (no
// gather / scatter launches around the solve):
//   front: y = blockdiag(M) * b[perm[0 .. nI)]   and   tail_out[i] = b[perm[nI + i]]            (i < ntail)
//   back : out[perm[k*rows + r]] (+)= z[k*zs + r] + alpha * M_k[r,:] . x[xoff[k] ..)   and
//          out[perm[nI + i]] (+)= x[i]                                                          (i < ntail)
// Same lane sums and reductions as dense_gemv_kernel / dense_gemv_windows_off_kernel.
__global__ void __launch_bounds__(kBlock) coarse_front_kernel(int64_t n, int64_t bs, const double *M, const double *b,
                                                              const int *perm, double *y, int64_t ntail, double *tail_out)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LMG_WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * kBlock / LMG_WAVE;
    for (int64_t row = wave; row < n; row += nwaves) {
        const double2 *M2 = reinterpret_cast<const double2 *>(M + row * bs);
        // a strip is a run of CONSECUTIVE unknowns (perm[k bs + c] = perm[k bs] + c: the caller's layout), so its
        // right-hand side is read in place; all loads of a chunk before its sums, same order of additions
        const double *x = b + perm[(row / bs) * bs];
        double s = 0.0;
        constexpr int CH = 8;
        for (int64_t j0 = lane; j0 < bs / 2; j0 += (int64_t)CH * LMG_WAVE) {
            double2 mv[CH];
            double xa[CH], xb[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int64_t j = j0 + (int64_t)c * LMG_WAVE;
                const int64_t jj = j < bs / 2 ? j : 0;
                mv[c] = M2[jj];
                xa[c] = x[2 * jj];
                xb[c] = x[2 * jj + 1];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (j0 + (int64_t)c * LMG_WAVE < bs / 2) {
                    s += mv[c].x * xa[c];
                    s += mv[c].y * xb[c];
                }
            }
        }
        s = lmg_wave_sum(s);
        if (lane == 0) y[row] = s;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ntail; i += (int64_t)gridDim.x * kBlock)
        tail_out[i] = b[perm[n + i]];
}

__global__ void __launch_bounds__(kBlock) coarse_back_kernel(int64_t nb, int64_t rows, int64_t cols, const double *M,
                                                             const double *x0, const int *xoff, const double *z0, int64_t zs,
                                                             double alpha, const int *perm, double *out, int accumulate,
                                                             int64_t ntail)
{
    const int lane = threadIdx.x & (LMG_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LMG_WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * kBlock / LMG_WAVE;
    for (int64_t row = wave; row < nb * rows; row += nwaves) {
        const int64_t k = row / rows, r = row - k * rows;
        const double2 *M2 = reinterpret_cast<const double2 *>(M + row * cols);
        const double *x = x0 + xoff[k];
        double s = 0.0;
        constexpr int CH = 4;                     // (all loads of a chunk before its sums, same order of additions)
        for (int64_t j0 = lane; j0 < cols / 2; j0 += (int64_t)CH * LMG_WAVE) {
            double2 mv[CH];
            double xa[CH], xb[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int64_t j = j0 + (int64_t)c * LMG_WAVE;
                const int64_t jj = j < cols / 2 ? j : 0;
                mv[c] = M2[jj];
                xa[c] = x[2 * jj];
                xb[c] = x[2 * jj + 1];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (j0 + (int64_t)c * LMG_WAVE < cols / 2) {
                    s += mv[c].x * xa[c];
                    s += mv[c].y * xb[c];
                }
            }
        }
        s = lmg_wave_sum(s);
        if (lane == 0) {
            const double v = z0[k * zs + r] + alpha * s;
            const int d = perm[row];
            out[d] = accumulate ? v + out[d] : v;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ntail; i += (int64_t)gridDim.x * kBlock) {
        const int d = perm[nb * rows + i];
        out[d] = accumulate ? x0[i] + out[d] : x0[i];
    }
}

// The same two products for blocks that are NOT runs of consecutive unknowns (coarse.py GridBlockSolver: rectangular
// blocks of a grid, padded to one size): every operand index comes from a table,
//   front: y[k*bs + r] = sum_c M_k[r][c] * b[idx[k*bs + c]]  (idx < 0: padding, contributes 0),  tail_out[i] = b[tail_idx[i]]
//   back : out[oidx[k*rows + r]] (+)= z[k*zs + r] + alpha * sum_c M_k[r][c] * x[xidx[k*cols + c]]  (oidx < 0: padding row),
//          out[tail_idx[i]] (+)= x[i]
// (fixed order of additions: lane sums over 128-entry passes, butterfly over the wave)
constexpr int kCoarseRows = 36;       // rows of a block per workgroup: 9 per wave, all their loads in flight at once

// One workgroup = kCoarseRows rows of ONE block: the block's operand segment (its gathered right-hand side / separator
// window) is gathered ONCE into LDS, then every wave walks its rows with all loads of all its rows issued before the first
// sum (rows are 50 - 300 entries: a wave per row with a grid-stride loop spent its time in three dependent round trips).
__global__ void __launch_bounds__(kBlock) coarse_front_gather_kernel(int64_t nblocks, int64_t bs, const double *M, const double *b,
                                                                     const int *idx, double *y, int64_t ntail,
                                                                     const int *tail_idx, double *tail_out)
{
    extern __shared__ double s_seg[];                              // bs doubles
    const int t = threadIdx.x, lane = t & (LMG_WAVE - 1), w = t / LMG_WAVE;
    const int64_t k = blockIdx.x, r0 = (int64_t)blockIdx.y * kCoarseRows;
    for (int64_t c = t; c < bs; c += kBlock) {
        const int g = idx[k * bs + c];
        s_seg[c] = g >= 0 ? b[g] : 0.0;
    }
    __syncthreads();
    const double2 *x2 = reinterpret_cast<const double2 *>(s_seg);
    constexpr int RPW = kCoarseRows / (kBlock / LMG_WAVE);         // 9 rows per wave
    const int64_t half = bs / 2;
    double tot[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) tot[q] = 0.0;
    for (int64_t j0 = 0; j0 < half; j0 += LMG_WAVE) {              // (rows longer than 128 entries: another pass)
        const int64_t j = j0 + lane;
        double2 mv[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int64_t r = r0 + w + 4 * q;
            mv[q] = (r < bs && j < half) ? reinterpret_cast<const double2 *>(M + (k * bs + r) * bs)[j] : double2{0.0, 0.0};
        }
        const double2 xv = j < half ? x2[j] : double2{0.0, 0.0};
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            double sacc = 0.0;
            sacc += mv[q].x * xv.x;
            sacc += mv[q].y * xv.y;
            tot[q] += lmg_wave_sum(sacc);
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t r = r0 + w + 4 * q;
        if (lane == 0 && r < bs) y[k * bs + r] = tot[q];
    }
    if (blockIdx.y == 0)
        for (int64_t i = k * kBlock + t; i < ntail; i += nblocks * kBlock) tail_out[i] = b[tail_idx[i]];
}

__global__ void __launch_bounds__(kBlock) coarse_back_gather_kernel(int64_t nb, int64_t rows, int64_t cols, const double *M,
                                                                    const double *x, const int *xidx, const double *z0,
                                                                    int64_t zs, double alpha, const int *oidx, double *out,
                                                                    int accumulate, int64_t ntail, const int *tail_idx)
{
    extern __shared__ double s_seg[];                              // cols doubles
    const int t = threadIdx.x, lane = t & (LMG_WAVE - 1), w = t / LMG_WAVE;
    const int64_t k = blockIdx.x, r0 = (int64_t)blockIdx.y * kCoarseRows;
    for (int64_t c = t; c < cols; c += kBlock) s_seg[c] = x[xidx[k * cols + c]];
    __syncthreads();
    const double2 *x2 = reinterpret_cast<const double2 *>(s_seg);
    constexpr int RPW = kCoarseRows / (kBlock / LMG_WAVE);
    const int64_t half = cols / 2;
    double tot[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) tot[q] = 0.0;
    for (int64_t j0 = 0; j0 < half; j0 += LMG_WAVE) {
        const int64_t j = j0 + lane;
        double2 mv[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int64_t r = r0 + w + 4 * q;
            mv[q] = (r < rows && j < half) ? reinterpret_cast<const double2 *>(M + (k * rows + r) * cols)[j] : double2{0.0, 0.0};
        }
        const double2 xv = j < half ? x2[j] : double2{0.0, 0.0};
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            double sacc = 0.0;
            sacc += mv[q].x * xv.x;
            sacc += mv[q].y * xv.y;
            tot[q] += lmg_wave_sum(sacc);
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t r = r0 + w + 4 * q;
        if (lane == 0 && r < rows) {
            const int d = oidx[k * rows + r];
            if (d >= 0) {
                const double v = z0[k * zs + r] + alpha * tot[q];
                out[d] = accumulate ? v + out[d] : v;
            }
        }
    }
    if (blockIdx.y == 0)
        for (int64_t i = k * kBlock + t; i < ntail; i += nb * kBlock) {
            const int d = tail_idx[i];
            out[d] = accumulate ? x[i] + out[d] : x[i];
        }
}

// How often every pattern id occurs on (even / odd line) x (even / odd column) of a grid with line stride W:
// counts[((y & 1) * 2 + (x & 1)) * 256 + id] (zeroed by the caller).  Setup helper of ops.ProlongTwin (a library
// histogram costs 0.4 s of code-object loading in a fresh process).
__global__ void __launch_bounds__(kBlock) pattern_parity_counts_kernel(int64_t n, int W, const unsigned char *pid, int *counts)
{
    __shared__ int s_cnt[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += kBlock) s_cnt[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const unsigned y = (unsigned)(i / W), x = (unsigned)(i - (int64_t)y * W);
        atomicAdd(&s_cnt[(((y & 1u) << 1) | (x & 1u)) * 256 + pid[i]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += kBlock)
        if (s_cnt[i]) atomicAdd(&counts[i], s_cnt[i]);
}

// dense[i][colidx[e]] += vals[e] for the entries e of row i (dense zero-initialised by the caller)
__global__ void __launch_bounds__(kBlock) csr_to_dense_kernel(int64_t n, int64_t m, const int *rowptr, const int *colidx,
                                                              const double *vals, double *dense)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) dense[i * m + colidx[e]] += vals[e];
}

// dst[k*ds + i] = src[k*ss + i], i < bs: strided copy of whole blocks
__global__ void __launch_bounds__(kBlock) block_copy_kernel(int64_t nb, int64_t bs, const double *src, int64_t ss,
                                                            double *dst, int64_t ds)
{
    const int64_t total = nb * bs;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t k = i / bs, r = i - k * bs;
        dst[k * ds + r] = src[k * ss + r];
    }
}

// Batched inverse of small dense matrices (n <= 128): one workgroup per matrix, the matrix in LDS,
// in-place Gauss-Jordan with partial pivoting (row swaps recorded, undone as column swaps at the end).
// Setup-time helper of the coarse solvers (the leaves of coarse.py's block Schur recursion): replaces
// thousands of 25-50 us rocSOLVER getf2 panel launches -- and the 0.4 s it takes to load that library
// in a fresh process -- by one launch.  info[k] = 1 when a pivot of matrix k is exactly zero.
constexpr int kInvMax = 128;
__global__ void __launch_bounds__(kBlock) batched_inverse_kernel(int n, const double *A, double *Ainv, int *info)
{
    extern __shared__ double s_a[];                  // n rows of stride n + 1
    __shared__ int s_piv[kInvMax];
    __shared__ int s_p;
    __shared__ double s_red_v[kBlock / LMG_WAVE];
    __shared__ int s_red_i[kBlock / LMG_WAVE];
    const int ld = n + 1, t = threadIdx.x;
    const double *Ak = A + (int64_t)blockIdx.x * n * n;
    for (int e = t; e < n * n; e += kBlock) s_a[(e / n) * ld + e % n] = Ak[e];
    __syncthreads();
    bool singular = false;
    for (int k = 0; k < n; ++k) {
        // pivot: largest |a[i][k]|, i >= k (ties: smallest i)
        double best = -1.0;
        int bi = k;
        for (int i = k + t; i < n; i += kBlock) {
            const double v = fabs(s_a[i * ld + k]);
            if (v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int off = LMG_WAVE / 2; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off, LMG_WAVE);
            const int oi = __shfl_down(bi, off, LMG_WAVE);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((t & (LMG_WAVE - 1)) == 0) { s_red_v[t / LMG_WAVE] = best; s_red_i[t / LMG_WAVE] = bi; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < kBlock / LMG_WAVE; ++w)
                if (s_red_v[w] > best || (s_red_v[w] == best && s_red_i[w] < bi)) { best = s_red_v[w]; bi = s_red_i[w]; }
            s_p = bi;
            s_piv[k] = bi;
        }
        __syncthreads();
        const int p = s_p;
        if (p != k)
            for (int j = t; j < n; j += kBlock) {
                const double tmp = s_a[k * ld + j];
                s_a[k * ld + j] = s_a[p * ld + j];
                s_a[p * ld + j] = tmp;
            }
        __syncthreads();
        const double piv = s_a[k * ld + k];
        if (piv == 0.0) { singular = true; break; }          // uniform: every thread reads the same LDS word
        __syncthreads();
        for (int j = t; j < n; j += kBlock) s_a[k * ld + j] = (j == k ? 1.0 : s_a[k * ld + j]) / piv;
        __syncthreads();
        // eliminate column k from every other row; a[i][k] becomes -f * (1 / piv)
        for (int e = t; e < n * n; e += kBlock) {
            const int i = e / n, j = e - i * n;
            if (i == k) continue;
            const double f = s_a[i * ld + k];
            if (j == k) continue;
            s_a[i * ld + j] -= f * s_a[k * ld + j];
        }
        __syncthreads();
        for (int i = t; i < n; i += kBlock)
            if (i != k) s_a[i * ld + k] = -s_a[i * ld + k] * s_a[k * ld + k];
        __syncthreads();
    }
    if (singular) {
        if (t == 0) info[blockIdx.x] = 1;
        return;
    }
    for (int k = n - 1; k >= 0; --k) {               // undo the row swaps: swap COLUMNS k and piv[k]
        const int p = s_piv[k];
        if (p != k)
            for (int i = t; i < n; i += kBlock) {
                const double tmp = s_a[i * ld + k];
                s_a[i * ld + k] = s_a[i * ld + p];
                s_a[i * ld + p] = tmp;
            }
        __syncthreads();
    }
    double *Ok = Ainv + (int64_t)blockIdx.x * n * n;
    for (int e = t; e < n * n; e += kBlock) Ok[e] = s_a[(e / n) * ld + e % n];
    if (t == 0) info[blockIdx.x] = 0;
}

// ---- exclusive scan (int32): per-block scan + block sums + add-back -------------------
constexpr int kScanBlock = 1024;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < LMG_WAVE; off <<= 1) {
        const int u = __shfl_up(v, off, LMG_WAVE);
        if (lane >= off) v += u;
    }
    return v;
}

// scans one tile; out[i+1] receives the inclusive value (caller pre-writes out[0] = 0)
__global__ void __launch_bounds__(kScanBlock) scan_tile_kernel(int64_t n, const int *in, int *out,
                                                               int *block_sums)
{
    __shared__ int s_wave[kScanBlock / LMG_WAVE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)t * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        sum += v[k];
    }
    const int incl = wave_incl_scan(sum, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int w = (lane < kScanBlock / LMG_WAVE) ? s_wave[lane] : 0;
        const int wi = wave_incl_scan(w, lane);
        if (lane < kScanBlock / LMG_WAVE) s_wave[lane] = wi - w;   // exclusive wave offsets
        if (lane == kScanBlock / LMG_WAVE - 1 && block_sums) block_sums[blockIdx.x] = wi;
    }
    __syncthreads();
    int run = s_wave[wave] + incl - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        run += v[k];
        if (base + k < n) out[base + k + 1] = run;
    }
    if (blockIdx.x == 0 && t == 0) out[0] = 0;
}

__global__ void __launch_bounds__(kScanBlock) scan_add_kernel(int64_t n, int *out, const int *block_offs)
{
    const int off = block_offs[blockIdx.x];
    if (off == 0) return;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    for (int i = threadIdx.x; i < kScanTile; i += kScanBlock)
        if (base + i < n) out[base + i + 1] += off;
}

int scan_rec(int64_t n, const int *in, int *out, int *scratch, hipStream_t st)
{
    const int64_t blocks = (n + kScanTile - 1) / kScanTile;
    if (blocks <= 1) {
        hipLaunchKernelGGL(scan_tile_kernel, dim3(1), dim3(kScanBlock), 0, st, n, in, out, (int *)nullptr);
        LMG_CHECK_LAUNCH();
        return LMG_OK;
    }
    int *sums = scratch;                 // blocks entries
    int *offs = scratch + blocks;        // blocks + 1 entries (exclusive scan of sums)
    int *rest = offs + blocks + 1;
    hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)blocks), dim3(kScanBlock), 0, st, n, in, out, sums);
    LMG_CHECK_LAUNCH();
    int rc = scan_rec(blocks, sums, offs, rest, st);
    if (rc != LMG_OK) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)blocks), dim3(kScanBlock), 0, st, n, out, offs);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // namespace

extern "C" {

int lmg_version(void) { return LMG_VERSION; }

const char *lmg_status_string(int s)
{
    switch (s) {
    case LMG_OK: return "ok";
    case LMG_ERR_ARG: return "invalid argument";
    case LMG_ERR_ALIGN: return "array base not 16-byte aligned";
    case LMG_ERR_LAUNCH: return "HIP launch/runtime error";
    case LMG_ERR_CAPACITY: return "row exceeds kernel capacity";
    case LMG_ERR_NODEVICE: return "no HIP device";
    default: return "unknown status";
    }
}

int lmg_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return LMG_ERR_NODEVICE;
    return c;
}

int lmg_tune_set(const char *key, int value)
{
    if (!key) return LMG_ERR_ARG;
    if (strcmp(key, "sweep_variant") == 0) return lmg_sweep_tune_set(value);
    if (strcmp(key, "pcsr_ju") == 0) return lmg_pcsr_tune_set(value);
    if (strcmp(key, "rpat_variant") == 0) return lmg_rpat_tune_set(value);
    if (strcmp(key, "rpat_nt_rows") == 0) return lmg_rpat_nt_set(value);
    if (strncmp(key, "stencil_", 8) == 0) return lmg_stencil_tune_set(key, value);
    if (strncmp(key, "fused_", 6) == 0) return lmg_fused_tune_set(key, value);
    if (strncmp(key, "tile_", 5) == 0) return lmg_tile_tune_set(key, value);
    if (strncmp(key, "dia_", 4) == 0) return lmg_dia_tune_set(key, value);
    if (strncmp(key, "sell_", 5) == 0) return lmg_sell_tune_set(key, value);
    if (strncmp(key, "gsw_", 4) == 0) return lmg_gsw_tune_set(key, value);
    if (strcmp(key, "gs_single_max") == 0) return lmg_gs_tune_set(value);
    return LMG_ERR_ARG;
}

int lmg_tune_get(const char *key)
{
    if (!key) return LMG_ERR_ARG;
    if (strcmp(key, "sweep_variant") == 0) return lmg_sweep_tune_get();
    if (strcmp(key, "pcsr_ju") == 0) return lmg_pcsr_tune_get();
    if (strcmp(key, "rpat_variant") == 0) return lmg_rpat_tune_get();
    if (strcmp(key, "rpat_nt_rows") == 0) return lmg_rpat_nt_get();
    if (strncmp(key, "stencil_", 8) == 0) return lmg_stencil_tune_get(key);
    if (strncmp(key, "fused_", 6) == 0) return lmg_fused_tune_get(key);
    if (strncmp(key, "tile_", 5) == 0) return lmg_tile_tune_get(key);
    if (strncmp(key, "dia_", 4) == 0) return lmg_dia_tune_get(key);
    if (strncmp(key, "sell_", 5) == 0) return lmg_sell_tune_get(key);
    if (strncmp(key, "gsw_", 4) == 0) return lmg_gsw_tune_get(key);
    if (strcmp(key, "gs_single_max") == 0) return lmg_gs_tune_get();
    return LMG_ERR_ARG;
}

int lmg_axpby(int64_t n, double alpha, const double *x, double beta, double *y, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !y))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!lmg_aligned16(x) || !lmg_aligned16(y)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, kBlock * 2)), dim3(kBlock), 0, lmg_stream(stream),
                       n, alpha, x, beta, y);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_vmul(int64_t n, double alpha, const double *x, const double *y, double *out, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !y || !out))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    hipLaunchKernelGGL(vmul_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, lmg_stream(stream), n, alpha, x, y, out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_copy(int64_t n, const double *src, double *dst, void *stream)
{
    if (n < 0 || (n > 0 && (!src || !dst))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice,
                       lmg_stream(stream)) != hipSuccess)
        return LMG_ERR_LAUNCH;
    return LMG_OK;
}

int lmg_zero(int64_t n, double *x, void *stream)
{
    if (n < 0 || (n > 0 && !x)) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (hipMemsetAsync(x, 0, (size_t)n * sizeof(double), lmg_stream(stream)) != hipSuccess)
        return LMG_ERR_LAUNCH;
    return LMG_OK;
}

int lmg_dot(int64_t n, const double *x, const double *y, double *partials, double *out, void *stream)
{
    if (n < 0 || !partials || !out || (n > 0 && (!x || !y))) return LMG_ERR_ARG;
    const unsigned grid = 1024;
    hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(kBlock), 0, lmg_stream(stream), n, x, y, partials);
    LMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(1024), 0, lmg_stream(stream), partials, (int)grid, out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_gather(int64_t n, const int32_t *idx, const double *x, double *buf, void *stream)
{
    if (n < 0 || (n > 0 && (!idx || !x || !buf))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, lmg_stream(stream), n, idx, x, buf);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_scatter(int64_t n, const int32_t *idx, const double *buf, double *x, void *stream)
{
    if (n < 0 || (n > 0 && (!idx || !x || !buf))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, lmg_stream(stream), n, idx, buf, x);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_dense_gemv(int64_t n, int64_t m, const double *M, const double *x, double *y, void *stream)
{
    if (n < 0 || m < 0 || (n > 0 && !y) || (n > 0 && m > 0 && (!M || !x))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!lmg_aligned16(M) || !lmg_aligned16(x)) return LMG_ERR_ALIGN;
    if (n <= 8192 && m >= 1024) {
        hipLaunchKernelGGL(dense_gemv_wgrow_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(kBlock), 0,
                           lmg_stream(stream), n, m, M, x, y);
        LMG_CHECK_LAUNCH();
        return LMG_OK;
    }
    hipLaunchKernelGGL(dense_gemv_kernel, dim3(grid_for(n, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), n, m, M, x, y, (int64_t)0);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_dense_gemv_blockdiag(int64_t nblocks, int64_t bs, const double *M, const double *x, double *y,
                             void *stream)
{
    if (nblocks < 0 || bs < 0 || (nblocks * bs > 0 && (!M || !x || !y)) || x == y) return LMG_ERR_ARG;
    if (nblocks * bs == 0) return LMG_OK;
    if (!lmg_aligned16(M) || !lmg_aligned16(x) || (bs % 2)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(dense_gemv_kernel, dim3(grid_for(nblocks * bs, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), nblocks * bs, bs, M, x, y, bs);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_dense_gemv_windows(int64_t nblocks, int64_t rows, int64_t cols, const double *M, const double *x,
                           int64_t x_stride, const double *z, int64_t z_stride, double alpha, double *y,
                           int64_t y_stride, void *stream)
{
    if (nblocks < 0 || rows < 0 || cols < 0 || x_stride < 0 || y_stride < rows || (z && z_stride < 0)) return LMG_ERR_ARG;
    if (nblocks * rows == 0) return LMG_OK;
    if (!M || !x || !y || x == y || z == y) return LMG_ERR_ARG;
    if (!lmg_aligned16(M) || !lmg_aligned16(x) || (cols % 2) || (x_stride % 2)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(dense_gemv_windows_kernel, dim3(grid_for(nblocks * rows, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), nblocks, rows, cols, M, x, x_stride, z, z_stride, alpha, y, y_stride);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_dense_gemv_windows_off(int64_t nblocks, int64_t rows, int64_t cols, const double *M, const double *x,
                               const int32_t *x_offsets, const double *z, int64_t z_stride, double alpha, double *y,
                               int64_t y_stride, void *stream)
{
    if (nblocks < 0 || rows < 0 || cols < 0 || y_stride < rows || (z && z_stride < 0)) return LMG_ERR_ARG;
    if (nblocks * rows == 0) return LMG_OK;
    if (!M || !x || !x_offsets || !y || x == y) return LMG_ERR_ARG;
    if (!lmg_aligned16(M) || (cols % 2)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(dense_gemv_windows_off_kernel, dim3(grid_for(nblocks * rows, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), nblocks, rows, cols, M, x, x_offsets, z, z_stride, alpha, y, y_stride);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_coarse_front(int64_t nblocks, int64_t bs, const double *M, const double *b, const int32_t *perm, double *y,
                     int64_t ntail, double *tail_out, void *stream)
{
    if (nblocks < 0 || bs < 0 || ntail < 0 || (nblocks * bs > 0 && (!M || !b || !perm || !y)) || (ntail > 0 && !tail_out) || b == y)
        return LMG_ERR_ARG;
    if (nblocks * bs == 0 && ntail == 0) return LMG_OK;
    if (!lmg_aligned16(M) || (bs % 2)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(coarse_front_kernel, dim3(grid_for(nblocks * bs > 0 ? nblocks * bs : 1, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), nblocks * bs, bs, M, b, perm, y, ntail, tail_out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_coarse_back(int64_t nblocks, int64_t rows, int64_t cols, const double *M, const double *x, const int32_t *x_offsets,
                    const double *z, int64_t z_stride, double alpha, const int32_t *perm, double *out, int accumulate,
                    int64_t ntail, void *stream)
{
    if (nblocks < 0 || rows < 0 || cols < 0 || ntail < 0 || z_stride < 0) return LMG_ERR_ARG;
    if (nblocks * rows == 0 && ntail == 0) return LMG_OK;
    if (!M || !x || !x_offsets || !z || !perm || !out || x == out || z == out) return LMG_ERR_ARG;
    if (!lmg_aligned16(M) || (cols % 2)) return LMG_ERR_ALIGN;
    hipLaunchKernelGGL(coarse_back_kernel, dim3(grid_for(nblocks * rows > 0 ? nblocks * rows : 1, kBlock / LMG_WAVE)), dim3(kBlock), 0,
                       lmg_stream(stream), nblocks, rows, cols, M, x, x_offsets, z, z_stride, alpha, perm, out, accumulate,
                       ntail);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_coarse_front_gather(int64_t nblocks, int64_t bs, const double *M, const double *b, const int32_t *idx, double *y,
                            int64_t ntail, const int32_t *tail_idx, double *tail_out, void *stream)
{
    if (nblocks < 0 || bs < 0 || ntail < 0 || (nblocks * bs > 0 && (!M || !b || !idx || !y)) ||
        (ntail > 0 && (!tail_out || !tail_idx || !b)) || b == y)
        return LMG_ERR_ARG;
    if (nblocks * bs == 0 && ntail == 0) return LMG_OK;
    if (!lmg_aligned16(M) || (bs % 2) || (reinterpret_cast<uintptr_t>(idx) & 7u)) return LMG_ERR_ALIGN;
    if (nblocks == 0 || bs == 0 || bs * 8 > 60000 || nblocks > 0x7fffffff) return LMG_ERR_CAPACITY;
    hipLaunchKernelGGL(coarse_front_gather_kernel, dim3((unsigned)nblocks, (unsigned)((bs + kCoarseRows - 1) / kCoarseRows)),
                       dim3(kBlock), (size_t)bs * 8, lmg_stream(stream), nblocks, bs, M, b, idx, y, ntail, tail_idx, tail_out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_coarse_back_gather(int64_t nblocks, int64_t rows, int64_t cols, const double *M, const double *x, const int32_t *xidx,
                           const double *z, int64_t z_stride, double alpha, const int32_t *oidx, double *out, int accumulate,
                           int64_t ntail, const int32_t *tail_idx, void *stream)
{
    if (nblocks < 0 || rows < 0 || cols < 0 || ntail < 0 || z_stride < 0) return LMG_ERR_ARG;
    if (nblocks * rows == 0 && ntail == 0) return LMG_OK;
    if (!M || !x || !xidx || !z || !oidx || !out || (ntail > 0 && !tail_idx) || x == out || z == out) return LMG_ERR_ARG;
    if (!lmg_aligned16(M) || (cols % 2) || (reinterpret_cast<uintptr_t>(xidx) & 7u)) return LMG_ERR_ALIGN;
    if (nblocks == 0 || rows == 0 || cols * 8 > 60000 || nblocks > 0x7fffffff) return LMG_ERR_CAPACITY;
    hipLaunchKernelGGL(coarse_back_gather_kernel, dim3((unsigned)nblocks, (unsigned)((rows + kCoarseRows - 1) / kCoarseRows)),
                       dim3(kBlock), (size_t)cols * 8, lmg_stream(stream), nblocks, rows, cols, M, x, xidx, z, z_stride, alpha, oidx,
                       out, accumulate, ntail, tail_idx);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_pattern_parity_counts(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t *counts, void *stream)
{
    if (n < 0 || line_stride < 1 || (n > 0 && (!pid || !counts))) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    hipLaunchKernelGGL(pattern_parity_counts_kernel, dim3(grid_for(n, kBlock * 16)), dim3(kBlock), 0, lmg_stream(stream), n,
                       line_stride, pid, counts);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_csr_to_dense(int64_t n, int64_t m, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                     double *dense, void *stream)
{
    if (n < 0 || m < 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!rowptr || !dense) return LMG_ERR_ARG;
    hipLaunchKernelGGL(csr_to_dense_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, lmg_stream(stream), n, m, rowptr,
                       colidx, vals, dense);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_batched_inverse(int64_t nmat, int32_t n, const double *A, double *Ainv, int32_t *info, void *stream)
{
    if (nmat < 0 || n < 1 || n > kInvMax) return LMG_ERR_ARG;
    if (nmat == 0) return LMG_OK;
    if (!A || !Ainv || !info || A == Ainv) return LMG_ERR_ARG;
    const size_t lds = (size_t)n * (n + 1) * sizeof(double);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(batched_inverse_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kInvMax * (kInvMax + 1) * sizeof(double)));
    hipLaunchKernelGGL(batched_inverse_kernel, dim3((unsigned)nmat), dim3(kBlock), lds, lmg_stream(stream), (int)n, A, Ainv,
                       info);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_block_copy(int64_t nblocks, int64_t bs, const double *src, int64_t src_stride, double *dst,
                   int64_t dst_stride, void *stream)
{
    if (nblocks < 0 || bs < 0 || src_stride < 0 || dst_stride < bs) return LMG_ERR_ARG;
    if (nblocks * bs == 0) return LMG_OK;
    if (!src || !dst) return LMG_ERR_ARG;
    hipLaunchKernelGGL(block_copy_kernel, dim3(grid_for(nblocks * bs, kBlock)), dim3(kBlock), 0, lmg_stream(stream),
                       nblocks, bs, src, src_stride, dst, dst_stride);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int64_t lmg_scan_scratch_count(int64_t n)
{
    int64_t total = 0;
    int64_t blocks = (n + kScanTile - 1) / kScanTile;
    while (blocks > 1) {
        total += 2 * blocks + 1;
        blocks = (blocks + kScanTile - 1) / kScanTile;
    }
    return total + 16;
}

int lmg_exclusive_scan_i32(int64_t n, const int32_t *in, int32_t *out, int32_t *scratch, void *stream)
{
    if (n < 0 || !out || (n > 0 && !in)) return LMG_ERR_ARG;
    if (n > kScanTile && !scratch) return LMG_ERR_ARG;
    return scan_rec(n, in, out, scratch, lmg_stream(stream));
}

int lmg_graph_begin(void *stream)
{
    if (hipStreamBeginCapture(lmg_stream(stream), hipStreamCaptureModeThreadLocal) != hipSuccess)
        return LMG_ERR_LAUNCH;
    return LMG_OK;
}

int lmg_graph_end(void *stream, void **exec_out)
{
    if (!exec_out) return LMG_ERR_ARG;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(lmg_stream(stream), &g) != hipSuccess || !g) return LMG_ERR_LAUNCH;
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return LMG_ERR_LAUNCH;
    *exec_out = ex;
    return LMG_OK;
}

int lmg_graph_launch(void *exec, void *stream)
{
    if (!exec) return LMG_ERR_ARG;
    if (hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(exec), lmg_stream(stream)) != hipSuccess)
        return LMG_ERR_LAUNCH;
    return LMG_OK;
}

int lmg_graph_destroy(void *exec)
{
    if (!exec) return LMG_OK;
    (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec));
    return LMG_OK;
}

}  // extern "C"
