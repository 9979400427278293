// Grid-stencil sweeps for gfx950: residual / Jacobi / SpMV on row-pattern matrices whose patterns
// are 3x3 stencils of ONE line stride W.
//
// A row-pattern matrix (rpat.hip: one uint8 pattern id per row, every distinct row stored once)
// qualifies when every entry of every pattern sits at  column - row = c*W + d,  c, d in {-1, 0, 1},
// with the entries of a pattern in ascending column order (sorted CSR rows): the 5-point operator
// of configs #2 / #4 (W = 4097: slots {-W, -1, 0, 1, W}), its 9-point Galerkin coarsenings, the
// tridiagonal 1-D operators (centre line only).  Nothing here knows about grid LINES: the format is
// expressed in linear offsets, identity rows at the boundary are just other patterns, and the host
// builder (ops.StencilTwin) derives W from the pattern table and refuses everything else.
//
// Why a second kernel for the same format: rpat.hip gathers x entry by entry, 8 bytes per lane --
// five vector-memory instructions per 64 rows for the 5-point operator, nine for the 9-point one,
// and those instructions (not their bytes, which hit L1 / L2) are what kept it at 86-90 % of the copy
// ceiling.  Here a lane owns TWO CONSECUTIVE rows r, r+1 and loads x[r-W .. r-W+1], x[r .. r+1],
// x[r+W .. r+W+1] with one 16-byte load each (8-byte aligned when W is odd: gfx950 global loads only
// need dword alignment); the left / right neighbours x[r-1], x[r+2] (and the diagonal ones of a 9-point
// row) come from the neighbouring lanes through DPP wave shifts, lanes 0 and 63 fetch theirs with one
// masked 8-byte load.  Per 128 rows: pattern ids (2 B/lane), b (16 B), three x windows (16 B), one
// edge load, one 16-byte store -- 7 vector-memory instructions instead of 16 (5-point) or 24.
//
// Arithmetic is unchanged: every row accumulates v*x over ITS pattern's entries in ascending column
// order = slot order, products and sums rounded separately, so the results are bit-identical to
// sweep.hip / pcsr.hip / rpat.hip and to the CPU oracle (tests assert array_equal).
#include <string.h>
#include "lmg_common.hpp"

namespace {

enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2 };

constexpr int kBlock = 256;
constexpr int kMaxPat = 64;       // stencil patterns held in LDS (9 values each)

typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) d2u { double a, b; };      // 16 bytes at 8-byte alignment

struct SArgs {
    int n;
    int W;
    int tiles;
    int tiles_per_xcd;
    int npat;
    unsigned umask;              // union of the slot masks of all patterns
    const unsigned char *pid;    // n
    const double *st_val;        // npat * 9, slot (c+1)*3 + (d+1)
    const int *st_mask;          // npat, bit s = slot s present
    const double *x;
    const double *b;
    double *out;
    double alpha, beta;
    double *partial;
};

__device__ __forceinline__ double dpp_from_lower_lane(double src, double lane0)
{
    // lane i <- lane i-1 (wave_shr:1); lane 0 keeps `lane0`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), __double2loint(src), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_upper_lane(double src, double lane63)
{
    // lane i <- lane i+1 (wave_shl:1); lane 63 keeps `lane63`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane63), __double2loint(src), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane63), __double2hiint(src), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// x[i], x[i+1] (0 where the index is outside [0, n)); ALIGNED16: i is even and the base 16-byte aligned
template <bool ALIGNED16, bool NT>
__device__ __forceinline__ d2 load_pair(const double *__restrict__ v, int64_t i, int n)
{
    d2 r;
    if (i >= 0 && i + 1 < n) {
        if (ALIGNED16) {
            const d2 *p = reinterpret_cast<const d2 *>(v + i);
            r = NT ? __builtin_nontemporal_load(p) : *p;
        } else {
            const d2u t = *reinterpret_cast<const d2u *>(v + i);
            r.x = t.a;
            r.y = t.b;
        }
    } else {
        r.x = (i >= 0 && i < n) ? v[i] : 0.0;
        r.y = (i + 1 >= 0 && i + 1 < n) ? v[i + 1] : 0.0;
    }
    return r;
}
__device__ __forceinline__ double load_one(const double *__restrict__ v, int64_t i, int n)
{
    return (i >= 0 && i < n) ? v[i] : 0.0;
}

template <bool DIAG>
struct Pair {
    int p2;           // pattern ids of rows r (low byte) and r+1
    d2 b, xc, xu, xd;
    double ec, eu, ed;     // lanes 0 / 63 only: x[r-1 (+-W)] resp. x[r+2 (+-W)]
};

// One pair of rows per lane and tile, no software prefetch: 50-63 registers, 8 waves per SIMD.  Measured
// on MI355X (tools/time_stencil.py, 4097^2 Jacobi): 0.0746 ms; two pairs per lane 0.0761; next tile's
// loads issued before the current one is processed (two register sets, 4-7 waves per SIMD) 0.0742 /
// 0.0974 ms, residual 0.0845 / 0.1007 -- occupancy hides the latency better than registers do.
template <int MODE, bool DIAG, bool NT>
__global__ void __launch_bounds__(kBlock, DIAG ? 4 : 8) stencil_sweep_kernel(SArgs a)
{
    constexpr int kP = 1;
    constexpr int kTileRows = 2 * kBlock * kP;
    __shared__ double s_val[kMaxPat * 9];
    __shared__ int s_mask[kMaxPat];
    __shared__ double s_rdiag[MODE == MODE_JACOBI ? kMaxPat : 1];
    __shared__ double s_red[kBlock / LMG_WAVE];

    const int t = threadIdx.x;
    const int lane = t & (LMG_WAVE - 1);
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int t_begin = xcd * a.tiles_per_xcd;
    const int t_end = min(a.tiles, t_begin + a.tiles_per_xcd);
    if (t_begin + slot >= t_end) {
        if (MODE == MODE_RESIDUAL && a.partial != nullptr && t == 0) a.partial[blockIdx.x] = 0.0;
        return;
    }

    const int n = a.n;
    const int64_t W = a.W;
    const bool use_u = (a.umask & 0x007u) != 0, use_d = (a.umask & 0x1C0u) != 0;
    const bool edge_lane = lane == 0 || lane == LMG_WAVE - 1;

    auto load_tile = [&](int tile, Pair<DIAG> (&P)[kP]) {
        const int tl = tile < t_end ? tile : t_end - 1;            // past the end: harmless re-read
#pragma unroll
        for (int q = 0; q < kP; ++q) {
            const int64_t r = (int64_t)tl * kTileRows + q * (2 * kBlock) + 2 * t;
            if (r + 1 < n) P[q].p2 = NT ? (int)__builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(a.pid + r))
                                        : (int)*reinterpret_cast<const unsigned short *>(a.pid + r);
            else P[q].p2 = r < n ? (int)a.pid[r] : 0;
            if (MODE != MODE_SPMV) P[q].b = load_pair<true, NT>(a.b, r, n);
            P[q].xc = load_pair<true, false>(a.x, r, n);
            const d2 zero2 = {0.0, 0.0};
            P[q].xu = use_u ? load_pair<false, false>(a.x, r - W, n) : zero2;
            P[q].xd = use_d ? load_pair<false, false>(a.x, r + W, n) : zero2;
            P[q].ec = P[q].eu = P[q].ed = 0.0;
            if (edge_lane) {
                const int64_t e = lane == 0 ? r - 1 : r + 2;
                P[q].ec = load_one(a.x, e, n);
                if (DIAG) {
                    P[q].eu = load_one(a.x, e - W, n);
                    P[q].ed = load_one(a.x, e + W, n);
                }
            }
        }
    };

    double local = 0.0;        // sum of r_i^2 over all rows of this workgroup (fixed order)
    auto process = [&](int tile, const Pair<DIAG> (&P)[kP]) {
#pragma unroll
        for (int q = 0; q < kP; ++q) {
            const int64_t r = (int64_t)tile * kTileRows + q * (2 * kBlock) + 2 * t;
            const int pA = P[q].p2 & 0xff, pB = (P[q].p2 >> 8) & 0xff;
            const int mA = s_mask[pA], mB = s_mask[pB];
            // window values: [0] = x[. - 1], [1] = x[.], [2] = x[. + 1], [3] = x[. + 2] relative to row r
            double wu[4], wc[4], wd[4];
            wc[1] = P[q].xc.x;
            wc[2] = P[q].xc.y;
            wc[0] = dpp_from_lower_lane(wc[2], P[q].ec);
            wc[3] = dpp_from_upper_lane(wc[1], P[q].ec);
            wu[1] = P[q].xu.x;
            wu[2] = P[q].xu.y;
            wd[1] = P[q].xd.x;
            wd[2] = P[q].xd.y;
            if (DIAG) {
                wu[0] = dpp_from_lower_lane(wu[2], P[q].eu);
                wu[3] = dpp_from_upper_lane(wu[1], P[q].eu);
                wd[0] = dpp_from_lower_lane(wd[2], P[q].ed);
                wd[3] = dpp_from_upper_lane(wd[1], P[q].ed);
            } else {
                wu[0] = wu[3] = wd[0] = wd[3] = 0.0;
            }
            double accA = 0.0, accB = 0.0;
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if (!DIAG && (s == 0 || s == 2 || s == 6 || s == 8)) continue;
                if (!((a.umask >> s) & 1u)) continue;                  // wave-uniform
                const int c = s / 3, d = s % 3;
                const double *w = c == 0 ? wu : (c == 1 ? wc : wd);
                const double vA = s_val[pA * 9 + s], vB = s_val[pB * 9 + s];
                const double tA = accA + vA * w[d], tB = accB + vB * w[d + 1];
                accA = ((mA >> s) & 1) ? tA : accA;
                accB = ((mB >> s) & 1) ? tB : accB;
            }
            double oA, oB;
            bool wrA = true, wrB = true;
            if (MODE == MODE_RESIDUAL) {
                oA = P[q].b.x - accA;
                oB = P[q].b.y - accB;
                if (r < n) local += oA * oA;
                if (r + 1 < n) local += oB * oB;
                wrA = wrB = a.out != nullptr;
            } else if (MODE == MODE_JACOBI) {
                const double rA = P[q].b.x - accA, rB = P[q].b.y - accB;
                oA = (mA >> 16) ? wc[1] : wc[1] + a.alpha * (s_rdiag[pA] * rA);
                oB = (mB >> 16) ? wc[2] : wc[2] + a.alpha * (s_rdiag[pB] * rB);
            } else {
                oA = accA;
                oB = accB;
                if (a.alpha != 1.0) {
                    oA = a.alpha * oA;
                    oB = a.alpha * oB;
                }
                if (a.beta != 0.0) {
                    const d2 y = load_pair<true, false>(a.out, r, n);
                    if (a.beta == 1.0) {
                        oA = y.x + oA;
                        oB = y.y + oB;
                    } else {
                        oA = a.beta * y.x + oA;
                        oB = a.beta * y.y + oB;
                    }
                }
            }
            if (wrA) {
                if (r + 1 < n) {
                    d2 o;
                    o.x = oA;
                    o.y = oB;
                    d2 *dst = reinterpret_cast<d2 *>(a.out + r);
                    if (NT) __builtin_nontemporal_store(o, dst);
                    else *dst = o;
                } else if (r < n) {
                    a.out[r] = oA;
                }
            }
        }
    };

    // The first tile's rows are requested BEFORE the pattern table is staged: on the small levels of a cycle a
    // workgroup has one tile, and table-then-rows would be two dependent trips to memory in a 5 us kernel.
    Pair<DIAG> PA[kP];
    int tile = t_begin + slot;
    load_tile(tile, PA);
    for (int i = t; i < a.npat * 9; i += kBlock) s_val[i] = a.st_val[i];
    for (int i = t; i < a.npat; i += kBlock) {
        const int m = a.st_mask[i];
        s_mask[i] = m;
        if (MODE == MODE_JACOBI) {
            const double d = (m & 16) ? a.st_val[i * 9 + 4] : 0.0;
            s_rdiag[i] = d != 0.0 ? 1.0 / d : 0.0;
            if (d == 0.0) s_mask[i] = m | (1 << 16);          // bit 16: no usable diagonal -> copy x
        }
    }
    __syncthreads();
    for (;;) {
        process(tile, PA);
        tile += nslots;
        if (tile >= t_end) break;
        load_tile(tile, PA);
    }
    if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
        // one partial per workgroup: the grid and the tile -> workgroup map are fixed for a given n,
        // so the norm is deterministic (and replayable from a hipGraph)
        const double tot = lmg_block_sum<kBlock>(local, s_red);
        if (t == 0) a.partial[blockIdx.x] = tot;
    }
}

__global__ void __launch_bounds__(1024) stencil_reduce_partials_kernel(const double *partial, int64_t count, double *out)
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

int g_stencil_nt_rows = 1 << 23;      // rows from which the id / b / out streams bypass the caches
int g_stencil_wgs_per_cu = 0;         // 0 = occupancy query

template <int MODE, bool DIAG, bool NT>
int launch_k(SArgs a, hipStream_t st)
{
    constexpr int kTileRows = 2 * kBlock;
    a.tiles = (a.n + kTileRows - 1) / kTileRows;
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stencil_sweep_kernel<MODE, DIAG, NT>, kBlock, 0) !=
            hipSuccess || per_cu < 1)
        per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    if (g_stencil_wgs_per_cu > 0 && g_stencil_wgs_per_cu < per_cu) per_cu = g_stencil_wgs_per_cu;
    int64_t grid = 256 * (int64_t)per_cu;
    if (grid > (int64_t)a.tiles_per_xcd * 8) grid = (int64_t)a.tiles_per_xcd * 8;
    hipLaunchKernelGGL((stencil_sweep_kernel<MODE, DIAG, NT>), dim3((unsigned)grid), dim3(kBlock), 0, st, a);
    LMG_CHECK_LAUNCH();
    return (int)grid;
}

template <int MODE>
int launch(SArgs a, hipStream_t st)
{
    const bool diag = (a.umask & 0x145u) != 0;                 // slots 0, 2, 6, 8
    const bool nt = a.n >= g_stencil_nt_rows;
    if (diag) return nt ? launch_k<MODE, true, true>(a, st) : launch_k<MODE, true, false>(a, st);
    return nt ? launch_k<MODE, false, true>(a, st) : launch_k<MODE, false, false>(a, st);
}

}  // namespace

int lmg_stencil_tune_set(const char *key, int v)
{
    if (strcmp(key, "stencil_nt_rows") == 0) {
        if (v < 1) return LMG_ERR_ARG;
        g_stencil_nt_rows = v;
        return LMG_OK;
    }
    if (strcmp(key, "stencil_wgs_per_cu") == 0) {
        if (v < 0 || v > 8) return LMG_ERR_ARG;
        g_stencil_wgs_per_cu = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_stencil_tune_get(const char *key)
{
    if (strcmp(key, "stencil_nt_rows") == 0) return g_stencil_nt_rows;
    if (strcmp(key, "stencil_wgs_per_cu") == 0) return g_stencil_wgs_per_cu;
    return LMG_ERR_ARG;
}

extern "C" {

int lmg_stencil_limits(int32_t *max_patterns)
{
    if (max_patterns) *max_patterns = kMaxPat;
    return LMG_OK;
}

int lmg_stencil_sweep(int mode, int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat,
                      const double *st_val, const int32_t *st_mask, uint32_t union_mask, const double *x,
                      const double *b, double *out, double alpha, double beta, double *partials, double *norm2,
                      void *stream)
{
    if (n < 0 || n >= INT32_MAX - 4096 || npat < 1 || npat > kMaxPat || (union_mask & ~0x1FFu)) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!pid || !st_val || !st_mask || !x) return LMG_ERR_ARG;
    // the line stride only matters when an upper / lower line is referenced at all
    if ((union_mask & 0x1C7u) && (line_stride < 3 || line_stride >= n)) return LMG_ERR_ARG;
    if (!lmg_aligned16(x) || (b && !lmg_aligned16(b)) || (out && !lmg_aligned16(out))) return LMG_ERR_ALIGN;
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
    a.W = line_stride;
    a.tiles = a.tiles_per_xcd = 0;
    a.npat = npat;
    a.umask = union_mask;
    a.pid = pid;
    a.st_val = st_val;
    a.st_mask = st_mask;
    a.x = x;
    a.b = b;
    a.out = out;
    a.alpha = alpha;
    a.beta = beta;
    a.partial = (mode == MODE_RESIDUAL) ? partials : nullptr;
    hipStream_t st = lmg_stream(stream);
    int nwg;                     // workgroups launched = partial sums written
    if (mode == MODE_RESIDUAL) nwg = launch<MODE_RESIDUAL>(a, st);
    else if (mode == MODE_JACOBI) nwg = launch<MODE_JACOBI>(a, st);
    else nwg = launch<MODE_SPMV>(a, st);
    if (nwg < 0) return nwg;
    if (mode == MODE_RESIDUAL && partials) {
        hipLaunchKernelGGL(stencil_reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partials, (int64_t)nwg, norm2);
        LMG_CHECK_LAUNCH();
    }
    return LMG_OK;
}

}  // extern "C"
