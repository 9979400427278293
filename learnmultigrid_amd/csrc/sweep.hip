// Fine-level CSR sweeps for gfx950: residual (+ fused ||r||^2), weighted Jacobi, SpMV.
//
// All three are HBM-bound (<= 0.25 flop/byte); no MFMA.  One skeleton:
//
//   * a workgroup owns a tile of BLOCK consecutive rows, hence ONE contiguous window
//     [rowptr[r0], rowptr[r1]) of colidx / vals, which it streams from HBM with
//     16-byte-per-lane coalesced loads (int4 / double2) into LDS tiles;
//   * then each thread walks its own row through the LDS tiles in storage order
//     (in-order accumulation, separate mul/add roundings -> bit-identical to SciPy's
//     csr_matvec); lane t handles row r0+t, so for stencil-like matrices the j-th
//     gather x[col] of 64 neighbouring rows is itself one coalesced 512-byte access
//     served by L1/L2;
//   * the diagonal for Jacobi is picked up while walking the row (no dinv array);
//   * blockIdx -> tile mapping is XCD-aware: block b runs on XCD b%8 (round-robin
//     dispatch), so XCD k is given the k-th contiguous eighth of the tiles and its
//     private 4 MiB L2 sees one sliding window of x instead of eight interleaved ones;
//   * rows longer than the LDS tile are handled by walking the window in CAP-sized
//     passes (running sums live in registers), so any CSR matrix is accepted;
//   * the tile geometry (BLOCK rows, CAP entries of LDS) is picked per launch from the
//     average row length so that a tile normally fits one pass while the LDS footprint
//     still allows 8 waves per SIMD (measured: occupancy is what hides the
//     rowptr -> matrix -> gather latency chain; 128 rows x 768 entries beats
//     256 x 2048 by 7 % on the 5-point fine level).
//
// Algorithmic HBM bytes per sweep: 12*nnz + 4*(n+1) + 24*n.
#include "lmg_common.hpp"

namespace {

enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2 };

struct SweepArgs {
    int n;
    int nnz;
    const int *rowptr;
    const int *colidx;
    const double *vals;
    const double *x;
    const double *b;   // rhs (residual, jacobi); unused for spmv
    double *out;       // r | x_out | y
    double alpha;      // omega (jacobi) | alpha (spmv)
    double beta;       // spmv only
    double *partial;   // residual: per-tile sum r^2, or nullptr
    int tiles;
    int tiles_per_xcd;
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// EXP (timing experiments only, results are wrong): 1 = no x gathers, 2 = no matrix stream
template <int MODE, int BLOCK, int CAP, bool NT, int EXP = 0>
__global__ void __launch_bounds__(BLOCK) csr_sweep_kernel(SweepArgs a)
{
    __shared__ __attribute__((aligned(16))) double s_val[CAP];
    __shared__ __attribute__((aligned(16))) int s_col[CAP];
    __shared__ double s_red[BLOCK / LMG_WAVE];

    const int t = threadIdx.x;
    const int tile = (int)(blockIdx.x & 7u) * a.tiles_per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= a.tiles) return;
    const int r0 = tile * BLOCK;
    const int r1 = min(a.n, r0 + BLOCK);
    // wave-uniform: scalar loads, issued before anything else
    const int base = a.rowptr[r0];
    const int end = a.rowptr[r1];

    const int row = r0 + t;
    int rs = 0, re = 0;
    double acc = 0.0, diag = 0.0, xi = 0.0, bv = 0.0;
    if (row < r1) {
        rs = a.rowptr[row];
        re = a.rowptr[row + 1];
        if (MODE != MODE_SPMV) bv = a.b[row];
    }

    const int a0 = base & ~3;   // 16-byte aligned start of the streamed window
    for (int w0 = a0; w0 < end; w0 += CAP) {
        if (w0 != a0) __syncthreads();          // previous pass fully consumed
#pragma unroll
        for (int i = t * 4; i < CAP; i += BLOCK * 4) {
            const int e = w0 + i;
            if (EXP == 2) {
                if (e < end) {
                    *reinterpret_cast<v4i *>(s_col + i) = v4i{r0, r0, r0, r0};
                    *reinterpret_cast<v2d *>(s_val + i) = v2d{1.0, 1.0};
                    *reinterpret_cast<v2d *>(s_val + i + 2) = v2d{1.0, 1.0};
                }
            } else if (e < end) {
                if (e + 3 < a.nnz) {
                    const v4i c = stream_load<NT>(reinterpret_cast<const v4i *>(a.colidx + e));
                    const v2d v0 = stream_load<NT>(reinterpret_cast<const v2d *>(a.vals + e));
                    const v2d v1 = stream_load<NT>(reinterpret_cast<const v2d *>(a.vals + e + 2));
                    *reinterpret_cast<v4i *>(s_col + i) = c;
                    *reinterpret_cast<v2d *>(s_val + i) = v0;
                    *reinterpret_cast<v2d *>(s_val + i + 2) = v1;
                } else {
                    for (int q = 0; q < 4; ++q)
                        if (e + q < a.nnz) {
                            s_col[i + q] = a.colidx[e + q];
                            s_val[i + q] = a.vals[e + q];
                        }
                }
            }
        }
        __syncthreads();
        const int lo = max(rs, w0) - w0;
        const int hi = min(re, w0 + CAP) - w0;
        for (int p = lo; p < hi; ++p) {
            const int c = s_col[p];
            const double v = s_val[p];
            const double xv = (EXP == 1) ? (double)c : a.x[EXP == 3 ? (c & 255) : (EXP == 4 ? (r0 + (c & 127)) : c)];
            acc += v * xv;
            if (MODE == MODE_JACOBI && c == row) {
                diag += v;
                xi = xv;
            }
        }
    }

    double local = 0.0;
    if (row < r1) {
        if (MODE == MODE_RESIDUAL) {
            const double r = bv - acc;
            if (a.out) a.out[row] = r;
            local = r * r;
        } else if (MODE == MODE_JACOBI) {
            const double r = bv - acc;
            if (diag != 0.0) {
                a.out[row] = xi + a.alpha * ((1.0 / diag) * r);
            } else {
                a.out[row] = a.x[row];
            }
        } else {
            double s = acc;
            if (a.alpha != 1.0) s = a.alpha * s;
            if (a.beta == 0.0) a.out[row] = s;
            else if (a.beta == 1.0) a.out[row] = a.out[row] + s;
            else a.out[row] = a.beta * a.out[row] + s;
        }
    }
    if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
        const double tot = lmg_block_sum<BLOCK>(local, s_red);
        if (t == 0) a.partial[tile] = tot;
    }
}

// ---- pipelined variant ------------------------------------------------------------------------
// Persistent workgroups (grid = a few per CU) that walk their XCD's tiles and keep the HBM
// stream in flight across tiles: while the waves of a workgroup wait for the x gathers of
// tile k, the matrix window of tile k+1 is already being loaded into registers
// (issue-early / write-late staging), and the rowptr scalars of tile k+2 are on their way.
// vmcnt retires in order, so the prefetch is issued AFTER the gathers of the first KU
// entries of each row: the gathers then complete without waiting for the prefetch.
// Arithmetic and accumulation order are identical to csr_sweep_kernel.
template <int MODE, int BLOCK, int CAP, int KU, int NBUF>
__global__ void __launch_bounds__(BLOCK) csr_sweep_pipe_kernel(SweepArgs a)
{
    constexpr int NL = (CAP + BLOCK * 4 - 1) / (BLOCK * 4);
    // two LDS tiles: with the tile of iteration k+1 going to the other buffer, ONE barrier
    // per tile is enough (a wave can only start overwriting buffer k%2 in iteration k+2,
    // i.e. after every wave has passed the barrier of k+1 and so finished reading tile k)
    // (NBUF = 1: single tile, second barrier at the end of every iteration, half the LDS)
    __shared__ __attribute__((aligned(16))) double s_val2[NBUF][CAP];
    __shared__ __attribute__((aligned(16))) int s_col2[NBUF][CAP];
    __shared__ double s_red[NBUF][BLOCK / LMG_WAVE];
    int buf = 0;
    double *s_val = s_val2[0];
    int *s_col = s_col2[0];

    const int t = threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int t_begin = xcd * a.tiles_per_xcd;
    const int t_end = min(a.tiles, t_begin + a.tiles_per_xcd);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    auto tile_lo = [&](int tl) { return a.rowptr[min(a.n, tl * BLOCK)]; };
    // scalars of this tile and of the next one (wave-uniform -> scalar loads)
    int base = tile_lo(tile), end = tile_lo(tile + 1);
    int nbase = 0, nend = 0;
    if (tile + nslots < t_end) {
        nbase = tile_lo(tile + nslots);
        nend = tile_lo(tile + nslots + 1);
    }

    // Register staging.  The loads are UNCONDITIONAL (lanes outside the window, and the one
    // partial 4-group at the very end of the matrix, read group 0 instead) so that the
    // compiler can count them: the gathers issued before a prefetch are then waited for
    // with vmcnt(#prefetch loads) instead of vmcnt(0).
    v4i pc[NL];
    v2d pv0[NL], pv1[NL];
    auto issue_matrix = [&](int w0, int w_end) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int e = w0 + t * 4 + l * BLOCK * 4;
            const bool ok = (t * 4 + l * BLOCK * 4 < CAP) && (e < w_end) && (e + 3 < a.nnz);
            const int ec = ok ? e : 0;
            pc[l] = *reinterpret_cast<const v4i *>(a.colidx + ec);
            pv0[l] = *reinterpret_cast<const v2d *>(a.vals + ec);
            pv1[l] = *reinterpret_cast<const v2d *>(a.vals + ec + 2);
        }
    };
    auto store_matrix = [&](int w0, int w_end) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int i = t * 4 + l * BLOCK * 4;
            const int e = w0 + i;
            if (i < CAP && e < w_end) {
                if (e + 3 < a.nnz) {
                    *reinterpret_cast<v4i *>(s_col + i) = pc[l];
                    *reinterpret_cast<v2d *>(s_val + i) = pv0[l];
                    *reinterpret_cast<v2d *>(s_val + i + 2) = pv1[l];
                } else {                                 // the matrix's final partial group
                    for (int q = 0; q < 4; ++q)
                        if (e + q < a.nnz) {
                            s_col[i + q] = a.colidx[e + q];
                            s_val[i + q] = a.vals[e + q];
                        }
                }
            }
        }
    };

    issue_matrix(base & ~3, end);

    while (true) {
        const int r0 = tile * BLOCK;
        const int r1 = min(a.n, r0 + BLOCK);
        const int row = r0 + t;
        const int a0 = base & ~3;
        int rs = 0, re = 0;
        double acc = 0.0, diag = 0.0, xi = 0.0, bv = 0.0;
        if (row < r1) {
            rs = a.rowptr[row];
            re = a.rowptr[row + 1];
            if (MODE != MODE_SPMV) bv = a.b[row];
        }
        store_matrix(a0, end);
        __syncthreads();

        const int nt = tile + nslots;
        const bool has_next = nt < t_end;
        int nnbase = 0, nnend = 0;

        const int lo = max(rs, a0) - a0;
        const int hi = min(re, a0 + CAP) - a0;
        {
            // first KU entries of the row: all gathers in flight before the prefetch is issued
            int c[KU];
            double v[KU], xg[KU];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const bool act = lo + k < hi;
                const int p = act ? lo + k : 0;
                const int cc = s_col[p];
                const double vv = s_val[p];
                c[k] = act ? cc : 0;
                v[k] = act ? vv : 0.0;
            }
#pragma unroll
            for (int k = 0; k < KU; ++k) xg[k] = a.x[c[k]];
            __builtin_amdgcn_sched_barrier(0);
            issue_matrix(has_next ? (nbase & ~3) : 0, has_next ? nend : 0);
            // scalars two tiles ahead -- issued here, not before the LDS reads above: scalar
            // loads share lgkmcnt with LDS and would stall the LDS-read -> gather chain
            if (nt + nslots < t_end) {
                nnbase = tile_lo(nt + nslots);
                nnend = tile_lo(nt + nslots + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const bool act = lo + k < hi;
                const double s2 = acc + v[k] * xg[k];
                acc = act ? s2 : acc;
                if (MODE == MODE_JACOBI) {
                    const bool dg = act && (c[k] == row);
                    diag = dg ? diag + v[k] : diag;
                    xi = dg ? xg[k] : xi;
                }
            }
        }
        for (int p = lo + KU; p < hi; ++p) {               // rows longer than KU (slow path)
            const int cc = s_col[p];
            const double vv = s_val[p];
            const double xv = a.x[cc];
            acc += vv * xv;
            if (MODE == MODE_JACOBI && cc == row) {
                diag += vv;
                xi = xv;
            }
        }
        // windows longer than the LDS tile: further passes, loaded synchronously
        for (int w0 = a0 + CAP; w0 < end; w0 += CAP) {
            __syncthreads();
            for (int i = t * 4; i < CAP; i += BLOCK * 4) {
                const int e = w0 + i;
                if (e < end) {
                    for (int q = 0; q < 4; ++q)
                        if (e + q < a.nnz) {
                            s_col[i + q] = a.colidx[e + q];
                            s_val[i + q] = a.vals[e + q];
                        }
                }
            }
            __syncthreads();
            const int l2 = max(rs, w0) - w0, h2 = min(re, w0 + CAP) - w0;
            for (int p = l2; p < h2; ++p) {
                const int cc = s_col[p];
                const double vv = s_val[p];
                const double xv = a.x[cc];
                acc += vv * xv;
                if (MODE == MODE_JACOBI && cc == row) {
                    diag += vv;
                    xi = xv;
                }
            }
        }

        double local = 0.0;
        if (row < r1) {
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
        if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
            const double tot = lmg_block_sum<BLOCK>(local, s_red[buf]);
            if (t == 0) a.partial[tile] = tot;
        }
        if (NBUF == 1 || end > a0 + CAP) __syncthreads();   // (extra passes wrote this buffer late)
        if (!has_next) break;
        buf = (NBUF == 2) ? (buf ^ 1) : 0;
        s_val = s_val2[buf];
        s_col = s_col2[buf];
        tile = nt;
        base = nbase;
        end = nend;
        nbase = nnbase;
        nend = nnend;
    }
}

// Final deterministic reduction of the per-tile partials: one workgroup, fixed order,
// four independent chains per thread so the loads pipeline.
__global__ void __launch_bounds__(1024) reduce_partials_kernel(const double *partial, int64_t count,
                                                               double *out)
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

// ---- tile geometry variants -----------------------------------------------------------------
// lmg_tune_set("sweep_variant", v): 0 = pick per launch from nnz/n (default); 1.. = forced.
struct Variant {
    int block, cap;
    bool nt;
};
constexpr int kNumVariants = 24;
constexpr Variant kVariants[kNumVariants] = {
    {0, 0, false},         //  0 auto
    {256, 2048, false},    //  1 (round-1 first cut)
    {256, 1536, false},    //  2
    {128, 768, false},     //  3 <= 5.3 nnz/row
    {128, 1024, false},    //  4 <= 7.1
    {128, 1536, false},    //  5 <= 10.7
    {128, 2048, false},    //  6 <= 14.2
    {64, 512, false},      //  7
    {64, 1024, false},     //  8 <= 14.2 (one-wave workgroups)
    {64, 2048, false},     //  9 <= 28.5, longer rows take several passes
    {512, 4096, false},    // 10
    {128, 768, true},      // 11 non-temporal matrix loads
    {256, 1536, true},     // 12
    {128, 768, false},     // 13 pipelined, KU 6, 12 workgroups / CU
    {256, 1536, false},    // 14 pipelined, KU 6,  6 workgroups / CU
    {128, 1536, false},    // 15 pipelined, KU 10
    {256, 2560, false},    // 16 pipelined, KU 10
    {128, 768, false},     // 17 pipelined, KU 6, 8 workgroups / CU
    {256, 1536, false},    // 18 pipelined, KU 6, 4 workgroups / CU
    {128, 768, false},     // 19 pipelined, KU 5, 12 workgroups / CU
    {128, 768, false},     // 20 pipelined, KU 5, 16 workgroups / CU
    {256, 1536, false},    // 21 pipelined, KU 5, 6 workgroups / CU
    {256, 1536, false},    // 22
    {128, 1536, false},    // 23 pipelined, KU 10, double-buffered
};
int g_sweep_variant = 0;

int pick_variant(int64_t n, int64_t nnz)
{
    if (g_sweep_variant != 0) return g_sweep_variant;
    // entries of a 128-row tile (+5 % for uneven rows, +8 for the aligned window start)
    const double need = (n > 0 ? (double)nnz / (double)n : 0.0) * 128 * 1.05 + 8;
    if (need <= 768) return 3;
    if (need <= 1024) return 4;
    if (need <= 1536) return 5;
    if (need <= 2048) return 6;
    if (need <= 2 * 1024) return 8;
    return 9;
}

template <int MODE, int BLOCK, int CAP, bool NT, int EXP = 0>
int launch_variant(SweepArgs a, hipStream_t st)
{
    a.tiles = (a.n + BLOCK - 1) / BLOCK;
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8);
    hipLaunchKernelGGL((csr_sweep_kernel<MODE, BLOCK, CAP, NT, EXP>), dim3(grid), dim3(BLOCK), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int MODE, int BLOCK, int CAP, int KU, int NBUF>
int launch_pipe(SweepArgs a, int wg_per_cu, hipStream_t st)
{
    if (a.nnz < 4) return launch_variant<MODE, BLOCK, CAP, false>(a, st);   // staging reads group 0
    a.tiles = (a.n + BLOCK - 1) / BLOCK;
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    int64_t grid = 256 * (int64_t)wg_per_cu;                 // 256 CUs, multiple of 8 (XCDs)
    if (grid > (int64_t)a.tiles_per_xcd * 8) grid = (int64_t)a.tiles_per_xcd * 8;
    hipLaunchKernelGGL((csr_sweep_pipe_kernel<MODE, BLOCK, CAP, KU, NBUF>), dim3((unsigned)grid),
                       dim3(BLOCK), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int MODE>
int launch_sweep(SweepArgs a, int variant, hipStream_t st)
{
    if (a.n == 0) return LMG_OK;
    switch (variant) {
    case 1: return launch_variant<MODE, 256, 2048, false>(a, st);
    case 2: return launch_variant<MODE, 256, 1536, false>(a, st);
    case 3: return launch_variant<MODE, 128, 768, false>(a, st);
    case 4: return launch_variant<MODE, 128, 1024, false>(a, st);
    case 5: return launch_variant<MODE, 128, 1536, false>(a, st);
    case 6: return launch_variant<MODE, 128, 2048, false>(a, st);
    case 7: return launch_variant<MODE, 64, 512, false>(a, st);
    case 8: return launch_variant<MODE, 64, 1024, false>(a, st);
    case 9: return launch_variant<MODE, 64, 2048, false>(a, st);
    case 10: return launch_variant<MODE, 512, 4096, false>(a, st);
    case 11: return launch_variant<MODE, 128, 768, true>(a, st);
    case 12: return launch_variant<MODE, 256, 1536, true>(a, st);
    case 13: return launch_pipe<MODE, 128, 768, 6, 2>(a, 12, st);
    case 14: return launch_pipe<MODE, 256, 1536, 6, 2>(a, 6, st);
    case 15: return launch_pipe<MODE, 128, 1536, 10, 1>(a, 8, st);
    case 16: return launch_pipe<MODE, 256, 2560, 10, 1>(a, 4, st);
    case 17: return launch_pipe<MODE, 128, 768, 6, 1>(a, 8, st);
    case 18: return launch_pipe<MODE, 256, 1536, 6, 1>(a, 4, st);
    case 19: return launch_pipe<MODE, 128, 768, 5, 2>(a, 12, st);
    case 20: return launch_pipe<MODE, 128, 768, 5, 1>(a, 12, st);
    case 21: return launch_pipe<MODE, 256, 1536, 5, 2>(a, 6, st);
    case 22: return launch_pipe<MODE, 256, 1536, 5, 1>(a, 6, st);
    case 23: return launch_pipe<MODE, 128, 1536, 10, 2>(a, 4, st);
    case 101: return launch_variant<MODE, 128, 768, false, 1>(a, st);   // timing experiments
    case 102: return launch_variant<MODE, 128, 768, false, 2>(a, st);
    case 103: return launch_variant<MODE, 128, 768, false, 3>(a, st);   // gathers hit 2 KB of x (L1)
    case 104: return launch_variant<MODE, 128, 768, false, 4>(a, st);   // gathers stay inside the tile's own rows
    default: return LMG_ERR_ARG;
    }
}

int check_csr(int64_t n, int64_t nnz, const void *rp, const void *ci, const void *va)
{
    if (n < 0 || nnz < 0 || n >= INT32_MAX || nnz >= INT32_MAX - 8192) return LMG_ERR_ARG;
    if (!rp || (nnz > 0 && (!ci || !va))) return LMG_ERR_ARG;
    if (!lmg_aligned16(ci) || !lmg_aligned16(va)) return LMG_ERR_ALIGN;
    return LMG_OK;
}

}  // namespace

int lmg_sweep_tune_set(int v)
{
    if ((v < 0 || v >= kNumVariants) && (v < 101 || v > 104)) return LMG_ERR_ARG;
    g_sweep_variant = v;
    return LMG_OK;
}
int lmg_sweep_tune_get(void) { return g_sweep_variant; }

extern "C" {

int64_t lmg_partials_count(int64_t n)
{
    // enough for the smallest tile of any variant (64 rows) and for lmg_dot's fixed 1024 partials
    int64_t t = (n + 63) / 64;
    return t < 1024 ? 1024 : t;
}

int lmg_csr_residual_norm2(int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci,
                           const double *va, const double *x, const double *b, double *r,
                           double *partials, double *norm2, void *stream)
{
    int st = check_csr(n, nnz, rp, ci, va);
    if (st != LMG_OK) return st;
    if (!x || !b) return LMG_ERR_ARG;
    if ((partials == nullptr) != (norm2 == nullptr)) return LMG_ERR_ARG;
    if (!r && !partials) return LMG_ERR_ARG;
    const int variant = pick_variant(n, nnz);
    SweepArgs a{(int)n, (int)nnz, rp, ci, va, x, b, r, 0.0, 0.0, partials, 0, 0};
    st = launch_sweep<MODE_RESIDUAL>(a, variant, lmg_stream(stream));
    if (st != LMG_OK) return st;
    if (partials) {
        const int rows = variant > 100 ? 128 : kVariants[variant].block;
        const int64_t tiles = (n + rows - 1) / rows;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, lmg_stream(stream),
                           partials, tiles, norm2);
        LMG_CHECK_LAUNCH();
    }
    return LMG_OK;
}

int lmg_csr_jacobi(int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci, const double *va,
                   const double *x_in, const double *b, double omega, double *x_out, void *stream)
{
    int st = check_csr(n, nnz, rp, ci, va);
    if (st != LMG_OK) return st;
    if (!x_in || !b || !x_out || x_in == x_out) return LMG_ERR_ARG;
    SweepArgs a{(int)n, (int)nnz, rp, ci, va, x_in, b, x_out, omega, 0.0, nullptr, 0, 0};
    return launch_sweep<MODE_JACOBI>(a, pick_variant(n, nnz), lmg_stream(stream));
}

int lmg_csr_spmv(int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci, const double *va,
                 const double *x, double *y, double alpha, double beta, void *stream)
{
    int st = check_csr(n, nnz, rp, ci, va);
    if (st != LMG_OK) return st;
    if (!y || (nnz > 0 && !x) || x == y) return LMG_ERR_ARG;
    SweepArgs a{(int)n, (int)nnz, rp, ci, va, x, nullptr, y, alpha, beta, nullptr, 0, 0};
    return launch_sweep<MODE_SPMV>(a, pick_variant(n, nnz), lmg_stream(stream));
}

}  // extern "C"
