// Packed-CSR sweeps for gfx950: the same residual / Jacobi / SpMV arithmetic as sweep.hip on a
// lossless re-encoding of the CSR matrix that moves fewer bytes through HBM.
//
// The sweeps are bandwidth-bound and sweep.hip already streams CSR at ~88 % of the copy
// ceiling of the chip, so the only lever left is the byte count.  PCSR keeps the CSR entry
// order (hence bit-identical row sums) and shrinks each stream where the data allows:
//
//   rowptr  (4 B/row)  -> rowlen  : uint8 per row + one int32 entry offset per 512-row tile
//   colidx  (4 B/nnz)  -> col     : uint16 offset from the tile's smallest column (COL16)
//                                   -- any matrix whose 512-row tiles span < 65536 columns,
//                                   i.e. every banded / grid matrix -- else int32 (COL32)
//   vals    (8 B/nnz)  -> val     : uint8 index into a dictionary of <= 256 distinct values
//                                   (VAL8, dictionary held in LDS), uint16 index into <= 65536
//                                   (VAL16, dictionary served by L2), else raw fp64 (VAL64)
//
// cfg#4 fine level (5-point, 3 distinct values): 88 B/DoF of CSR -> 40 B/DoF.  Matrices with
// all-distinct values (jittered meshes, learned Q) still drop from 12 to 10 B/nnz.
//
// Kernel: persistent workgroups of 128 threads, 4 rows per thread (tile = 512 rows: one
// scalar-load chain, one row-length scan and two barriers per 512 rows instead of per 128),
// XCD-aware tile ownership, packed streams staged into LDS still packed (3 B/entry), rows
// walked in storage order by lane t = row t (mod 128) so the x gathers of a wave stay
// coalesced.  Same in-order, FMA-free accumulation as everywhere else.
#include "lmg_common.hpp"

namespace {

enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2 };
enum { COL16 = 0, COL32 = 1 };
enum { VAL8 = 0, VAL16 = 1, VAL64 = 2 };

// Tile geometries (threads per workgroup, rows per thread): 512-row tiles for short rows,
// 128- and 64-row tiles when the rows are long enough that 512 of them would not fit the
// LDS budget (25-entry rows of L2-type Galerkin operators).  Chosen at pack time.
constexpr int kDefaultTileRows = 512;

struct PArgs {
    int n;
    int nnz;
    int tiles;
    int tiles_per_xcd;
    int cap;                 // LDS capacity in entries (>= largest tile + alignment slack)
    int tile_rows;           // 512, 128 or 64
    const int *tile_base;    // tiles + 1 entry offsets
    const int *tile_colbase; // tiles
    const unsigned char *rowlen;
    const void *col;
    const void *val;
    const double *dict;
    int ndict;
    const double *x;
    const double *b;
    double *out;
    double alpha, beta;
    double *partial;
};

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < LMG_WAVE; off <<= 1) {
        const int u = __shfl_up(v, off, LMG_WAVE);
        if (lane >= off) v += u;
    }
    return v;
}

template <int COLMODE> struct ColT { typedef unsigned short type; };
template <> struct ColT<COL32> { typedef int type; };
template <int VALMODE> struct ValT { typedef unsigned char type; };
template <> struct ValT<VAL16> { typedef unsigned short type; };
template <> struct ValT<VAL64> { typedef double type; };

// copy the 16-byte-aligned byte window [lo16, hi) of a global array into LDS, 16 B per lane
template <int kBlock>
__device__ __forceinline__ void stage_bytes(const unsigned char *g, long lo16, long hi, unsigned char *s, int t)
{
    for (long off = lo16 + (long)t * 16; off < hi; off += (long)kBlock * 16)
        *reinterpret_cast<v4u *>(s + (off - lo16)) = *reinterpret_cast<const v4u *>(g + off);
}

// JU = entries of each row handled per step (JU x 4 independent gathers in flight per thread)
// EXP (timing experiments only, wrong results): 1 = no x gathers, 2 = no gathers and no staging
#ifndef LMG_PCSR_WAVES
#define LMG_PCSR_WAVES 1
#endif
template <int MODE, int COLMODE, int VALMODE, int JU, int EXP = 0, int kBlock = 128, int kRpt = 4>
__global__ void __launch_bounds__(kBlock, LMG_PCSR_WAVES) pcsr_sweep_kernel(PArgs a)
{
    constexpr int kTileRows = kBlock * kRpt;
    constexpr int NW = kBlock / LMG_WAVE;
    typedef typename ColT<COLMODE>::type col_t;
    typedef typename ValT<VALMODE>::type val_t;
    constexpr int CPER = 16 / (int)sizeof(col_t);   // entries per 16-byte chunk
    constexpr int VPER = 16 / (int)sizeof(val_t);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [dict (VAL8: 2048 B)] [col bytes] [val bytes] [scan: 8 ints] [red: 2 doubles]
    double *s_dict = reinterpret_cast<double *>(smem);
    const int dict_bytes = (VALMODE == VAL8) ? 4096 : 0;     // dictionary + its reciprocals
    double *s_rdict = s_dict + 256;
    const int col_bytes = ((a.cap * (int)sizeof(col_t) + 15) & ~15) + 16;
    const int val_bytes = ((a.cap * (int)sizeof(val_t) + 15) & ~15) + 16;
    unsigned char *s_colb = smem + dict_bytes;
    unsigned char *s_valb = s_colb + col_bytes;
    int *s_scan = reinterpret_cast<int *>(s_valb + val_bytes);
    double *s_red = reinterpret_cast<double *>(s_scan + 8);     // s_scan: kRpt*NW <= 8 ints
    const col_t *s_col = reinterpret_cast<const col_t *>(s_colb);
    const val_t *s_val = reinterpret_cast<const val_t *>(s_valb);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int t_begin = xcd * a.tiles_per_xcd;
    const int t_end = min(a.tiles, t_begin + a.tiles_per_xcd);
    if (t_begin + slot >= t_end) return;

    if (VALMODE == VAL8) {
        for (int i = t; i < 256; i += kBlock) {
            const double d = i < a.ndict ? a.dict[i] : 0.0;
            s_dict[i] = d;
            // Jacobi divides by the diagonal, which is a dictionary entry: one division per
            // dictionary entry and workgroup instead of one per row (same IEEE division)
            s_rdict[i] = (MODE == MODE_JACOBI && d != 0.0) ? 1.0 / d : 0.0;
        }
    }

    for (int tile = t_begin + slot; tile < t_end; tile += nslots) {
        const int base = a.tile_base[tile];
        const int end = a.tile_base[tile + 1];
        const int cb0 = (COLMODE == COL16) ? a.tile_colbase[tile] : 0;
        const int r0 = tile * kTileRows;

        // ---- row lengths and their exclusive scan in row order (row = r0 + k*128 + t) ----
        int len[kRpt], incl[kRpt];
        double bv[kRpt];
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            const int row = r0 + k * kBlock + t;
            len[k] = row < a.n ? (int)a.rowlen[row] : 0;
            bv[k] = (MODE != MODE_SPMV && row < a.n) ? a.b[row] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            incl[k] = wave_incl_scan_i(len[k], lane);
            if (lane == 63) s_scan[k * NW + wave] = incl[k];
        }
        // ---- stage the packed streams (still packed) ---------------------------------------
        const int ac = base & ~(CPER - 1);          // first staged entry of each stream
        const int av = base & ~(VPER - 1);
        if (EXP != 2) {
            stage_bytes<kBlock>(reinterpret_cast<const unsigned char *>(a.col), (long)ac * sizeof(col_t),
                        (long)end * sizeof(col_t), s_colb, t);
            stage_bytes<kBlock>(reinterpret_cast<const unsigned char *>(a.val), (long)av * sizeof(val_t),
                        (long)end * sizeof(val_t), s_valb, t);
        }
        __syncthreads();

        // entry offset of each of this thread's 4 rows (k-major row order inside the tile)
        int rs[kRpt];
        {
            int run = 0;
#pragma unroll
            for (int k = 0; k < kRpt; ++k) {
                int before = run;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int sw = s_scan[k * NW + w];
                    if (w < wave) before += sw;
                    run += sw;
                }
                rs[k] = base + before + incl[k] - len[k];
            }
        }
        // walk the 4 rows TOGETHER, entry j of each per step: the 4 gathers of a step are
        // independent, so one L2 round trip serves 4 rows (each row still accumulates its own
        // entries in storage order).  Inactive slots read the tile's first entry (valid).
        double acc[kRpt], diag[kRpt], xi[kRpt];
        int didx[kRpt];                    // VAL8: dictionary index of the (single) diagonal entry
        int maxlen = 0;
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            acc[k] = 0.0;
            diag[k] = 0.0;
            xi[k] = 0.0;
            didx[k] = -1;
            maxlen = max(maxlen, len[k]);
        }
        for (int j0 = 0; j0 < maxlen; j0 += JU) {
            int c[JU][kRpt], vi[JU][kRpt];
            double v[JU][kRpt], xv[JU][kRpt];
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) {
                    const int p = (j0 + jj < len[k]) ? rs[k] + j0 + jj : base;
                    c[jj][k] = cb0 + (int)s_col[p - ac];
                    vi[jj][k] = 0;
                    if constexpr (VALMODE == VAL8) {
                        vi[jj][k] = s_val[p - av];
                        v[jj][k] = s_dict[vi[jj][k]];
                    } else if constexpr (VALMODE == VAL16) v[jj][k] = a.dict[s_val[p - av]];
                    else v[jj][k] = s_val[p - av];
                }
            }
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) xv[jj][k] = EXP ? (double)c[jj][k] : a.x[c[jj][k]];
            }
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) {
                    const bool act = j0 + jj < len[k];
                    const double s2 = acc[k] + v[jj][k] * xv[jj][k];
                    acc[k] = act ? s2 : acc[k];
                    if (MODE == MODE_JACOBI) {
                        const bool dg = act && (c[jj][k] == r0 + k * kBlock + t);
                        diag[k] = dg ? diag[k] + v[jj][k] : diag[k];
                        xi[k] = dg ? xv[jj][k] : xi[k];
                        if (VALMODE == VAL8) didx[k] = dg ? (didx[k] == -1 ? vi[jj][k] : -2) : didx[k];
                    }
                }
            }
        }
        double local = 0.0;
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            const int row = r0 + k * kBlock + t;
            if (row < a.n) {
                if (MODE == MODE_RESIDUAL) {
                    const double r = bv[k] - acc[k];
                    if (a.out) a.out[row] = r;
                    local += r * r;
                } else if (MODE == MODE_JACOBI) {
                    const double r = bv[k] - acc[k];
                    if (diag[k] != 0.0) {
                        double inv;
                        if (VALMODE == VAL8 && didx[k] >= 0) inv = s_rdict[didx[k]];
                        else inv = 1.0 / diag[k];
                        a.out[row] = xi[k] + a.alpha * (inv * r);
                    } else {
                        a.out[row] = a.x[row];
                    }
                } else {
                    double s = acc[k];
                    if (a.alpha != 1.0) s = a.alpha * s;
                    if (a.beta == 0.0) a.out[row] = s;
                    else if (a.beta == 1.0) a.out[row] = a.out[row] + s;
                    else a.out[row] = a.beta * a.out[row] + s;
                }
            }
        }
        if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
            const double tot = lmg_block_sum<kBlock>(local, s_red);
            if (t == 0) a.partial[tile] = tot;
        }
        __syncthreads();          // LDS tiles, s_scan and s_red are reused by the next tile
    }
}

__global__ void __launch_bounds__(1024) pcsr_reduce_partials_kernel(const double *partial, int64_t count,
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

constexpr int kMaxLds = 64 * 1024;

int lds_bytes(int cap, int colmode, int valmode)
{
    const int cs = colmode == COL16 ? 2 : 4;
    const int vs = valmode == VAL8 ? 1 : (valmode == VAL16 ? 2 : 8);
    return (valmode == VAL8 ? 4096 : 0) + ((cap * cs + 15) & ~15) + 16 + ((cap * vs + 15) & ~15) + 16 + 32 + 16;
}

int g_pcsr_ju = 0;      // 0 = pick from the average row length; 1, 2, 3, 5 = forced (tuning)

template <int MODE, int COLMODE, int VALMODE, int JU>
int launch_ju(PArgs a, hipStream_t st);
template <int MODE, int COLMODE, int VALMODE, int JU, int kBlock, int kRpt>
int launch_geo(PArgs a, hipStream_t st);

template <int MODE, int COLMODE, int VALMODE>
int launch(PArgs a, hipStream_t st)
{
    int ju = g_pcsr_ju;
    if (ju == 0) {
        // measured on MI355X (tools/tune_sweep.py): short rows want the smallest register
        // footprint (JU 1); ~9-entry rows want 3 steps of 3, or 2 steps of 5 when the gathers
        // go to a much longer vector (restriction: SpMV mode)
        const double avg = (double)a.nnz / (double)a.n;
        ju = avg <= 6.0 ? 1 : ((MODE == MODE_SPMV || avg > 12.0) ? 5 : 3);
    }
    if (ju == 101 || ju == 102) {
        if (COLMODE != COL16 || VALMODE != VAL8 || a.tile_rows != 512) return LMG_ERR_ARG;
        constexpr int kBlock = 128;
        const int lds = lds_bytes(a.cap, COLMODE, VALMODE);
        int64_t grid = 256 * 10;
        if (grid > (int64_t)a.tiles_per_xcd * 8) grid = (int64_t)a.tiles_per_xcd * 8;
        if (ju == 101)
            hipLaunchKernelGGL((pcsr_sweep_kernel<MODE, COL16, VAL8, 1, 1>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
        else
            hipLaunchKernelGGL((pcsr_sweep_kernel<MODE, COL16, VAL8, 1, 2>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
        LMG_CHECK_LAUNCH();
        return LMG_OK;
    }
    if (a.tile_rows != 512 && ju < 3) ju = 3;
    switch (ju) {
    case 1: return launch_ju<MODE, COLMODE, VALMODE, 1>(a, st);
    case 5: return launch_ju<MODE, COLMODE, VALMODE, 5>(a, st);
    default: return launch_ju<MODE, COLMODE, VALMODE, 3>(a, st);
    }
}

template <int MODE, int COLMODE, int VALMODE, int JU>
int launch_ju(PArgs a, hipStream_t st)
{
    // instantiated combinations: 512-row tiles with JU 1/3/5 and every encoding; 128- and
    // 64-row tiles (long rows) with JU 3/5 and VAL8 / VAL64 only
    if (a.tile_rows == 512) return launch_geo<MODE, COLMODE, VALMODE, JU, 128, 4>(a, st);
    if constexpr (JU >= 3 && VALMODE != VAL16) {
        if (a.tile_rows == 128) return launch_geo<MODE, COLMODE, VALMODE, JU, 128, 1>(a, st);
        if (a.tile_rows == 64) return launch_geo<MODE, COLMODE, VALMODE, JU, 64, 1>(a, st);
    }
    return LMG_ERR_ARG;
}

template <int MODE, int COLMODE, int VALMODE, int JU, int kBlock, int kRpt>
int launch_geo(PArgs a, hipStream_t st)
{
    const int lds = lds_bytes(a.cap, COLMODE, VALMODE);
    if (lds > kMaxLds) return LMG_ERR_CAPACITY;
    // persistent grid: exactly as many workgroups per CU as registers + LDS admit
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, pcsr_sweep_kernel<MODE, COLMODE, VALMODE, JU, 0, kBlock, kRpt>, kBlock, (size_t)lds) != hipSuccess ||
        per_cu < 1)
        per_cu = 4;
    if (per_cu > 32) per_cu = 32;
    int64_t grid = 256 * (int64_t)per_cu;
    if (grid > (int64_t)a.tiles_per_xcd * 8) grid = (int64_t)a.tiles_per_xcd * 8;
    hipLaunchKernelGGL((pcsr_sweep_kernel<MODE, COLMODE, VALMODE, JU, 0, kBlock, kRpt>), dim3((unsigned)grid),
                       dim3(kBlock), lds, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int MODE>
int dispatch(PArgs a, int colmode, int valmode, hipStream_t st)
{
    if (colmode == COL16) {
        if (valmode == VAL8) return launch<MODE, COL16, VAL8>(a, st);
        if (valmode == VAL16) return launch<MODE, COL16, VAL16>(a, st);
        if (valmode == VAL64) return launch<MODE, COL16, VAL64>(a, st);
    } else if (colmode == COL32) {
        if (valmode == VAL8) return launch<MODE, COL32, VAL8>(a, st);
        if (valmode == VAL16) return launch<MODE, COL32, VAL16>(a, st);
        if (valmode == VAL64) return launch<MODE, COL32, VAL64>(a, st);
    }
    return LMG_ERR_ARG;
}

}  // namespace

int lmg_pcsr_tune_set(int ju)
{
    if (ju != 0 && ju != 1 && ju != 3 && ju != 5 && ju != 101 && ju != 102) return LMG_ERR_ARG;
    g_pcsr_ju = ju;
    return LMG_OK;
}
int lmg_pcsr_tune_get(void) { return g_pcsr_ju; }

extern "C" {

int lmg_pcsr_tile_rows(void) { return kDefaultTileRows; }

int lmg_pcsr_sweep(int mode, int64_t n, int64_t nnz, int32_t tile_rows, int32_t tile_cap, const int32_t *tile_base,
                   const int32_t *tile_colbase, const uint8_t *rowlen, const void *col, int colmode,
                   const void *val, int valmode, const double *dict, int32_t ndict, const double *x,
                   const double *b, double *out, double alpha, double beta, double *partials,
                   double *norm2, void *stream)
{
    if (n < 0 || nnz < 0 || n >= INT32_MAX || nnz >= INT32_MAX - 65536 || tile_cap < 0) return LMG_ERR_ARG;
    if (tile_rows != 512 && tile_rows != 128 && tile_rows != 64) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!tile_base || !rowlen || (nnz > 0 && (!col || !val || !x))) return LMG_ERR_ARG;
    if (colmode == COL16 && !tile_colbase) return LMG_ERR_ARG;
    if (valmode != VAL64 && (!dict || ndict <= 0)) return LMG_ERR_ARG;
    if ((valmode == VAL8 && ndict > 256) || (valmode == VAL16 && ndict > 65536)) return LMG_ERR_ARG;
    if (!lmg_aligned16(col) || !lmg_aligned16(val)) return LMG_ERR_ALIGN;
    if (mode == MODE_SPMV) {
        if (!out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_JACOBI) {
        if (!b || !out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_RESIDUAL) {
        if (!b || (partials == nullptr) != (norm2 == nullptr) || (!out && !partials)) return LMG_ERR_ARG;
    } else {
        return LMG_ERR_ARG;
    }
    PArgs a;
    a.n = (int)n;
    a.nnz = (int)nnz;
    a.tile_rows = tile_rows;
    a.tiles = (int)((n + tile_rows - 1) / tile_rows);
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    a.cap = tile_cap + 32;
    a.tile_base = tile_base;
    a.tile_colbase = tile_colbase;
    a.rowlen = rowlen;
    a.col = col;
    a.val = val;
    a.dict = dict;
    a.ndict = ndict;
    a.x = x;
    a.b = b;
    a.out = out;
    a.alpha = alpha;
    a.beta = beta;
    a.partial = (mode == MODE_RESIDUAL) ? partials : nullptr;
    hipStream_t st = lmg_stream(stream);
    int rc;
    if (mode == MODE_RESIDUAL) rc = dispatch<MODE_RESIDUAL>(a, colmode, valmode, st);
    else if (mode == MODE_JACOBI) rc = dispatch<MODE_JACOBI>(a, colmode, valmode, st);
    else rc = dispatch<MODE_SPMV>(a, colmode, valmode, st);
    if (rc != LMG_OK) return rc;
    if (mode == MODE_RESIDUAL && partials) {
        hipLaunchKernelGGL(pcsr_reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partials, (int64_t)a.tiles, norm2);
        LMG_CHECK_LAUNCH();
    }
    return LMG_OK;
}

}  // extern "C"
