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

template <int MODE, int BLOCK, int CAP>
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
            if (e < end) {
                if (e + 3 < a.nnz) {
                    const v4i c = *(reinterpret_cast<const v4i *>(a.colidx + e));
                    const v2d v0 = *(reinterpret_cast<const v2d *>(a.vals + e));
                    const v2d v1 = *(reinterpret_cast<const v2d *>(a.vals + e + 2));
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
            const double xv = a.x[c];
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
// lmg_tune_set("sweep_variant", v): 0 = pick per launch from nnz/n (default); 1..6 = forced.  (Round 1
// carried 24 variants incl. pipelined and non-temporal ones; none of them beat these on any matrix class --
// DESIGN.md section 4 -- and the lossless twins took over the hot path, so they were removed.)
struct Variant {
    int block, cap;
};
constexpr int kNumVariants = 7;
constexpr Variant kVariants[kNumVariants] = {
    {0, 0},          // 0 auto
    {128, 768},      // 1 <= 5.3 nnz/row
    {128, 1024},     // 2 <= 7.1
    {128, 1536},     // 3 <= 10.7
    {128, 2048},     // 4 <= 14.2
    {64, 1024},      // 5 <= 14.2 (one-wave workgroups)
    {64, 2048},      // 6 <= 28.5, longer rows take several passes
};
int g_sweep_variant = 0;

int pick_variant(int64_t n, int64_t nnz)
{
    if (g_sweep_variant != 0) return g_sweep_variant;
    // entries of a 128-row tile (+5 % for uneven rows, +8 for the aligned window start)
    const double need = (n > 0 ? (double)nnz / (double)n : 0.0) * 128 * 1.05 + 8;
    if (need <= 768) return 1;
    if (need <= 1024) return 2;
    if (need <= 1536) return 3;
    if (need <= 2048) return 4;
    return 6;
}

template <int MODE, int BLOCK, int CAP>
int launch_variant(SweepArgs a, hipStream_t st)
{
    a.tiles = (a.n + BLOCK - 1) / BLOCK;
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    const unsigned grid = (unsigned)(a.tiles_per_xcd * 8);
    hipLaunchKernelGGL((csr_sweep_kernel<MODE, BLOCK, CAP>), dim3(grid), dim3(BLOCK), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int MODE>
int launch_sweep(SweepArgs a, int variant, hipStream_t st)
{
    if (a.n == 0) return LMG_OK;
    switch (variant) {
    case 1: return launch_variant<MODE, 128, 768>(a, st);
    case 2: return launch_variant<MODE, 128, 1024>(a, st);
    case 3: return launch_variant<MODE, 128, 1536>(a, st);
    case 4: return launch_variant<MODE, 128, 2048>(a, st);
    case 5: return launch_variant<MODE, 64, 1024>(a, st);
    case 6: return launch_variant<MODE, 64, 2048>(a, st);
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
    if (v < 0 || v >= kNumVariants) return LMG_ERR_ARG;
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
        const int rows = kVariants[variant].block;
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
