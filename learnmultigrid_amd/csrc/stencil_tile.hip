// Fused smoothing passes for SMALL grid-stencil levels (format of stencil.hip) for gfx950:
//     x_out = J^S(x_in)   [r = b - A x_out]          S = 1..3 weighted-Jacobi sweeps, x_in == NULL: zero iterate
// -- what stencil_fused.hip does for the multi-million-row levels, with the iterates in LDS instead of registers.
// On a level of 10^4 .. 10^6 rows a sweep is a 2 - 7 us launch (tools/time_small.py) and the register kernel is no
// help: a wave there walks 12 - 40 lines one after the other, 2 100 cycles each.  Here a WORKGROUP owns a tile of
// 64 columns x RR lines (its inner 64 - 2H x RR - 2H part is what it stores; H = S, +1 with the residual, -1 from a
// zero iterate), loads it once -- x, b, pattern ids --, runs the S sweeps between two LDS buffers with all four
// waves working on different lines of the same sweep, and stores the inner part: S (+1) launches become one, the
// dependent chain is S + 2 barriers long instead of S (+1) kernel boundaries.
// The arithmetic per row and sweep is the instruction sequence of stencil_sweep_kernel (slot order, separate
// multiply and add, omega * (rdiag * r)), so every value is bit-identical to the separate launches; as in
// stencil_fused.hip everything is a LINEAR index i = line * W + column, a column outside [0, W) being the linear
// neighbour in the adjacent line.
// (Tried: two columns per lane, 128-column tiles -- half the per-element overhead on paper, slower in practice: 9-point
// 2049^2 61.5 / 48.0 us against 49.6 / 41.4 us for 3 sweeps + residual / 3 sweeps; twice the LDS per tile, half the waves.)
#include <string.h>
#include "lmg_common.hpp"

namespace {

constexpr int kRB = 2;                        // consecutive lines of a tile a wave owns: a tile of RR lines is RR / 2 waves
                                              // (4 lines per wave: cfg#2 cycle 0.131 instead of 0.122 ms, cfg#4 equal)
constexpr int kMaxPat = 64;
constexpr int kCols = 64;                     // columns of a tile = lanes of a wave
constexpr int kLS = kCols + 2;                // LDS line stride: one guard column on either side
constexpr unsigned kMask5 = 0x0BAu;
constexpr unsigned kMask9 = 0x1FFu;

__device__ __forceinline__ double dpp_lower(double src)      // lane i <- lane i-1, lane 0 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_upper(double src)      // lane i <- lane i+1, lane 63 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct __attribute__((aligned(8))) d2u { double a, b; };

struct TArgs {
    int n, W, lines, npat;
    int tiles_x, tiles_y;
    const unsigned char *pid;
    const double *st_val;
    const int *st_mask;
    const double *x;          // may be NULL with ZERO
    const double *b;
    double *out;
    double *r;                // may be NULL without RESID
    double omega;
    int hot;
    double hot_val[9];
    double hot_rdiag;
    // PROL: the sweeps start from x + P e_c (Multigrid.py:115 folded into the post-smoothing pass), formed when the tile
    // is loaded: row i = y * W + x of P reads e_c at ((y >> 1) * Wc + (x >> 1)) + {0, 1, Wc, Wc + 1} (slots 0..3)
    const double *ec;
    int nc, Wc;
    const unsigned char *ppid;
    const double *pp_val;     // [pp_npat][4]
    const int *pp_mask;       // [pp_npat]
    int pp_npat;
    // REST: b_coarse = R (b - A x_out) instead of the residual (Multigrid.py:90 + :93 folded into the pre-smoothing pass):
    // row (Y, X) of R reads r at (2Y * W + 2X) + c * W + d, c, d in {-1, 0, 1} (slots 0..8); nc, Wc as above
    double *bc;
    const unsigned char *rpid;
    const double *rp_val;     // [rp_npat][9]
    const int *rp_mask;       // [rp_npat]
    int rp_npat;
};

template <int S, unsigned UM, bool RESID, bool ZERO, int RR, bool PROL = false, bool REST = false, int RBV = kRB>
// (16-wave workgroups: at most 64 VGPRs, so that two of them share a CU -- the variants with the restriction had 65 - 67)
__global__ void __launch_bounds__(RR / RBV * LMG_WAVE, (RR / RBV == 16 && !PROL) ? 8 : 1) stencil_tile_kernel(TArgs a)
{
    static_assert(!PROL || (!RESID && !ZERO), "the correction is folded into post-smoothing passes only");
    static_assert(!REST || (RESID && !PROL), "the restriction replaces the store of the residual");
    // halo: one more with REST -- the residual has to be exact one line / column beyond the stored part
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0) + (REST ? 1 : 0);
    constexpr int kWaves = RR / RBV, kBlock = kWaves * LMG_WAVE;
    static_assert(RR > 2 * H + 1 && kCols > 2 * H && RR % RBV == 0, "tile smaller than its halo / lines per wave");
    // LDS holds the two iterate buffers only: right-hand side and pattern ids of a wave's own lines never change and
    // stay in its registers (32-line tiles: 39 KB instead of 60, i.e. four workgroups per CU instead of two).
    __shared__ double s_x[2][RR * kLS];
    __shared__ double s_val[kMaxPat * 9];
    __shared__ int s_mask[kMaxPat];
    __shared__ double s_rdiag[kMaxPat];
    __shared__ double s_pv[PROL ? kMaxPat * 4 : 1];
    __shared__ int s_pm[PROL ? kMaxPat : 1];
    __shared__ double s_rv[REST ? kMaxPat * 9 : 1];
    __shared__ int s_rm[REST ? kMaxPat : 1];

    const int t = threadIdx.x, lane = t & (LMG_WAVE - 1), wave = t >> 6;
    const int tx = (int)blockIdx.x % a.tiles_x, ty = (int)blockIdx.x / a.tiles_x;
    const int c0 = tx * (kCols - 2 * H) - H, y0 = ty * (RR - 2 * H) - H;
    const int n = a.n;
    const int64_t W = a.W;
    constexpr int RB = RBV;                                      // consecutive lines of the tile a wave owns
    constexpr bool DIAG = (UM & 0x145u) != 0;
    const int rb0 = wave * RB;

    // ---- the wave's lines are requested first, the pattern table is staged while they are in flight -------------
    double lx[RB], bk[RB];
    int pk[RB];                                                   // pattern id | 0x100 where the element is a row of the matrix
    double le[PROL ? RB : 1][4];                                  // PROL: the 2 x 2 coarse window of every element
    int lq[(PROL || REST) ? RB : 1];                              //       and its pattern id in P / REST: the id of R's row
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int y = y0 + rb0 + k;
        const int64_t i = (int64_t)y * W + c0 + lane;
        const bool ok = y >= 0 && y < a.lines && i >= 0 && i < n;
        const int64_t j = ok ? i : 0;
        lx[k] = (!ZERO && ok) ? a.x[j] : 0.0;
        bk[k] = ok ? a.b[j] : 0.0;
        pk[k] = ok ? ((int)a.pid[j] | 0x100) : 0;
        if (PROL) {
            // the element is row j of P whatever its place in the tile (a column outside [0, W) is an element of the
            // adjacent line): its own line and column decide the window.  On grids at least two tiles wide one step to
            // the neighbouring line is enough; narrower ones divide.
            const int c = c0 + lane;
            int yy, xx;
            if (a.W >= 2 * kCols) {                               // uniform
                yy = c < 0 ? y - 1 : (c >= a.W ? y + 1 : y);
                xx = c < 0 ? c + a.W : (c >= a.W ? c - a.W : c);
            } else {
                yy = (int)((unsigned)j / (unsigned)a.W);
                xx = (int)((unsigned)j - (unsigned)yy * (unsigned)a.W);
            }
            const int64_t base = (int64_t)(yy >> 1) * a.Wc + (xx >> 1);
            lq[k] = ok ? (int)a.ppid[j] : 0;
            // slots 0, 1 and 2, 3 are neighbours in memory: two 16-byte loads (8-byte aligned is enough); slots a
            // pattern does not have may point anywhere inside the vector
            const int64_t b0 = ok ? min(base, (int64_t)a.nc - 2) : 0, b1 = ok ? min(base + a.Wc, (int64_t)a.nc - 2) : 0;
            const d2u e01 = *reinterpret_cast<const d2u *>(a.ec + b0), e23 = *reinterpret_cast<const d2u *>(a.ec + b1);
            // (a window clamped at the end of the vector starts one element early: its first slot is the second value)
            le[k][0] = b0 == base ? e01.a : e01.b;
            le[k][1] = e01.b;
            le[k][2] = b1 == base + a.Wc ? e23.a : e23.b;
            le[k][3] = e23.b;
        }
        if (REST) {
            // elements on (even line, even column) of the grid carry a row of R
            const int c = c0 + lane;
            const bool crow = ok && !(y & 1) && !(c & 1) && c >= 0 && c < W;
            const int64_t jc = (int64_t)(y >> 1) * a.Wc + (c >> 1);
            lq[k] = crow ? ((int)a.rpid[jc < a.nc ? jc : 0] | 0x100) : 0;
        }
    }
    for (int i = t; i < a.npat * 9; i += kBlock) s_val[i] = a.st_val[i];
    for (int i = t; i < a.npat; i += kBlock) {
        const int m = a.st_mask[i];
        const double dg = (m & 16) ? a.st_val[i * 9 + 4] : 0.0;
        s_rdiag[i] = dg != 0.0 ? 1.0 / dg : 0.0;
        s_mask[i] = dg == 0.0 ? (m | (1 << 16)) : m;             // bit 16: no usable diagonal -> copy x
    }
    for (int i = t; i < 2 * RR; i += kBlock) {                    // guard columns of both iterate buffers
        const int r = i >> 1, g = (i & 1) ? kLS - 1 : 0;
        s_x[0][r * kLS + g] = 0.0;
        s_x[1][r * kLS + g] = 0.0;
    }
    if (PROL) {
        for (int i = t; i < a.pp_npat * 4; i += kBlock) s_pv[i] = a.pp_val[i];
        for (int i = t; i < a.pp_npat; i += kBlock) s_pm[i] = a.pp_mask[i];
        __syncthreads();
        // x + P e: the sums of lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 1) in the same order
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int m = s_pm[lq[k]];
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double tv = acc + s_pv[lq[k] * 4 + q] * le[k][q];
                acc = ((m >> q) & 1) ? tv : acc;
            }
            lx[k] = (pk[k] >> 8) ? lx[k] + acc : 0.0;
        }
    }
    if (REST) {
        for (int i = t; i < a.rp_npat * 9; i += kBlock) s_rv[i] = a.rp_val[i];
        for (int i = t; i < a.rp_npat; i += kBlock) s_rm[i] = a.rp_mask[i];
    }
    const int hot = a.hot >= 0 ? (a.hot | 0x100) : -1;
    bool mine = true;
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        s_x[0][(rb0 + k) * kLS + 1 + lane] = lx[k];
        s_x[1][(rb0 + k) * kLS + 1 + lane] = 0.0;
        mine = mine && (pk[k] == hot || !(pk[k] >> 8));          // (elements outside the matrix are not stored)
    }
    // every lane of every line of this wave holds the frequent pattern: its values sit in scalar registers
    const bool all_hot = __all(mine);
    __syncthreads();

    const double omega = a.omega;
    double hv[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) hv[s] = a.hot_val[s];
    const double hrd = a.hot_rdiag;

    // A wave slides a three-line window down its lines: the centre values of a line come from LDS (one 8-byte read
    // per lane), its left / right neighbours from the neighbouring lanes (DPP; lanes 0 / 63 get the zero of the
    // guard columns -- they are halo), so a line costs one LDS read instead of nine.
    struct Win { double m, c, p; };
    auto line = [&](const double *src, int r, bool sides) -> Win {
        Win w;
        w.c = src[r * kLS + 1 + lane];
        w.m = sides ? dpp_lower(w.c) : 0.0;
        w.p = sides ? dpp_upper(w.c) : 0.0;
        return w;
    };
    // A x of the centre line of (u, c, d) for this lane, slot order = column order
    auto apply = [&](const Win &u, const Win &c, const Win &d, int p, bool hotp) -> double {
        const double w[9] = {u.m, u.c, u.p, c.m, c.c, c.p, d.m, d.c, d.p};
        double acc = 0.0;
        if (hotp) {
#pragma unroll
            for (int s = 0; s < 9; ++s)
                if ((UM >> s) & 1u) acc = acc + hv[s] * w[s];
        } else {
            const int q = p & 0xff, m = s_mask[q];
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if (!((UM >> s) & 1u)) continue;
                const double tv = acc + s_val[q * 9 + s] * w[s];
                acc = ((m >> s) & 1) ? tv : acc;
            }
        }
        return acc;
    };
    // The wave's block of RB lines, straight-line: all its lines are read first, then either the frequent pattern
    // everywhere or every line through the pattern table -- no branch inside either path, so the LDS latencies and
    // the dependent sums of the RB lines overlap.  Lines outside [lo, hi) are computed from clamped (meaningless)
    // neighbours and not kept.
    auto block = [&](const double *src, int lo, int hi, auto &&emit) {
        Win ln[RB + 2];
#pragma unroll
        for (int j = 0; j < RB + 2; ++j) {
            const int rr = min(max(rb0 - 1 + j, 0), RR - 1);
            ln[j] = line(src, rr, DIAG || (j >= 1 && j <= RB));
        }
        if (all_hot) {                                             // wave-uniform
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const double acc = apply(ln[k], ln[k + 1], ln[k + 2], pk[k], true);
                emit(k, rb0 + k >= lo && rb0 + k < hi, true, ln[k + 1].c, acc);
            }
        } else {
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const double acc = apply(ln[k], ln[k + 1], ln[k + 2], pk[k], false);
                emit(k, rb0 + k >= lo && rb0 + k < hi, false, ln[k + 1].c, acc);
            }
        }
    };

    // ---- the sweeps: iterate s goes from buffer (s - 1) & 1 to buffer s & 1 -------------------------------------
#pragma unroll
    for (int s = 1; s <= S; ++s) {
        const double *src = s_x[(s - 1) & 1];
        double *dst = s_x[s & 1];
        if (ZERO && s == 1) {
            // first sweep from a zero iterate: x = omega * (D^-1 b) on every line (lmg_vmul's bits)
#pragma unroll
            for (int k = 0; k < RB; ++k)
                dst[(rb0 + k) * kLS + 1 + lane] = (pk[k] >> 8) ? omega * (s_rdiag[pk[k] & 0xff] * bk[k]) : 0.0;
        } else {
            // (lines 0 and RR - 1 have no line above / below)
            block(src, 1, RR - 1, [&](int k, bool keep, bool hotp, double xc, double acc) {
                const double res = bk[k] - acc;
                double nx;
                if (hotp) {
                    nx = xc + omega * (hrd * res);
                } else {
                    const int q = pk[k] & 0xff;
                    nx = (s_mask[q] >> 16) ? xc : xc + omega * (s_rdiag[q] * res);
                }
                if (keep) dst[(rb0 + k) * kLS + 1 + lane] = (pk[k] >> 8) ? nx : 0.0;
            });
        }
        __syncthreads();
    }

    // ---- outputs: the inner part of the tile, inside the line, rows of the matrix --------------------------------
    const double *fin = s_x[S & 1];
    const bool col_ok = lane >= H && lane < kCols - H && c0 + lane >= 0 && c0 + lane < W;
    if (REST) {
        // the residual goes to the other LDS buffer (all lines but the first and the last: exact where it is read),
        // then every element that carries a row of R sums its nine entries in column order -- the sums of
        // lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 0)
        double *rl = s_x[(S + 1) & 1];
        block(fin, 1, RR - 1, [&](int k, bool keep, bool, double xc, double acc) {
            const int r = rb0 + k;
            if (keep) rl[r * kLS + 1 + lane] = bk[k] - acc;
            if (r >= H && r < RR - H && col_ok && (pk[k] >> 8)) a.out[(int64_t)(y0 + r) * W + c0 + lane] = xc;
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int r = rb0 + k;
            const int q = lq[k] & 0xff, m = s_rm[q];
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const int rr = min(max(r + e / 3 - 1, 0), RR - 1);
                const double tv = acc + s_rv[q * 9 + e] * rl[rr * kLS + 1 + lane + e % 3 - 1];
                acc = ((m >> e) & 1) ? tv : acc;
            }
            if ((lq[k] >> 8) && r >= H && r < RR - H && col_ok)
                a.bc[(int64_t)((y0 + r) >> 1) * a.Wc + ((c0 + lane) >> 1)] = acc;
        }
    } else if (RESID) {
        block(fin, H, RR - H, [&](int k, bool keep, bool, double xc, double acc) {
            const int64_t i = (int64_t)(y0 + rb0 + k) * W + c0 + lane;
            if (keep && col_ok && (pk[k] >> 8)) {
                a.out[i] = xc;
                a.r[i] = bk[k] - acc;
            }
        });
    } else {
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int r = rb0 + k;
            if (r >= H && r < RR - H && col_ok && (pk[k] >> 8))
                a.out[(int64_t)(y0 + r) * W + c0 + lane] = fin[r * kLS + 1 + lane];
        }
    }
}

int g_tile_rows = 16;       // lines per tile (16, 32; 0 = 32) on grids of fewer than g_tile_big_lines lines: more, smaller workgroups
                            // where a level is a handful of tiles (cfg#4 cycle 0.4725 -> 0.4687 ms, 1025^2 / 4 levels 0.1150 -> 0.1103,
                            // cfg#2 0.0728 -> 0.0706)
int g_tile_prol_wide_lines = 768;    // grids of at least this many lines: the pass with the correction on 8-wave workgroups
int g_tile_rows_big = 0;    // the same for grids of at least g_tile_big_lines lines
int g_tile_big_lines = 600;

template <int S, unsigned UM, bool RESID, bool ZERO, int RR, bool PROL = false, bool REST = false, int RBV = kRB>
int launch5(TArgs a, hipStream_t st)
{
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0) + (REST ? 1 : 0);
    a.tiles_x = (a.W + (kCols - 2 * H) - 1) / (kCols - 2 * H);
    a.tiles_y = (a.lines + (RR - 2 * H) - 1) / (RR - 2 * H);
    const int64_t grid = (int64_t)a.tiles_x * a.tiles_y;
    if (grid > 0x7fffffff) return LMG_ERR_CAPACITY;
    hipLaunchKernelGGL((stencil_tile_kernel<S, UM, RESID, ZERO, RR, PROL, REST, RBV>), dim3((unsigned)grid), dim3(RR / RBV * LMG_WAVE), 0, st,
                       a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

static int tile_rows_for(const TArgs &a)
{
    const int rr = a.lines >= g_tile_big_lines ? g_tile_rows_big : g_tile_rows;
    return rr == 0 ? 32 : rr;     // measured in the cycle (cfg#4): 0.672 ms with 32-line tiles (16 waves), 0.688 with 16 (8 waves)
}

template <int S, unsigned UM, bool RESID, bool ZERO, bool PROL = false, bool REST = false>
int launch4(TArgs a, hipStream_t st)
{
    if (tile_rows_for(a) == 16) return launch5<S, UM, RESID, ZERO, 16, PROL, REST>(a, st);
    // the pass with the correction needs 81 VGPRs: a 16-wave workgroup then fills a CU alone; on levels with many tiles
    // it runs 8 waves of four lines each (2049^2, 9-point: 69 instead of 81 us)
    if constexpr (PROL) {
        if (a.lines >= g_tile_prol_wide_lines) return launch5<S, UM, RESID, ZERO, 32, PROL, REST, 4>(a, st);
    }
    return launch5<S, UM, RESID, ZERO, 32, PROL, REST>(a, st);
}

template <unsigned UM>
int launch_prol(TArgs a, int sweeps, hipStream_t st)
{
    switch (sweeps) {
    case 1: return launch4<1, UM, false, false, true>(a, st);
    case 2: return launch4<2, UM, false, false, true>(a, st);
    default: return launch4<3, UM, false, false, true>(a, st);
    }
}

template <unsigned UM>
int launch_rest(TArgs a, int sweeps, bool zero, hipStream_t st)
{
    switch (sweeps) {
    case 1: return zero ? launch4<1, UM, true, true, false, true>(a, st) : launch4<1, UM, true, false, false, true>(a, st);
    case 2: return zero ? launch4<2, UM, true, true, false, true>(a, st) : launch4<2, UM, true, false, false, true>(a, st);
    default: return zero ? launch4<3, UM, true, true, false, true>(a, st) : launch4<3, UM, true, false, false, true>(a, st);
    }
}

template <int S, unsigned UM>
int launch2(TArgs a, bool resid, bool zero, hipStream_t st)
{
    if (resid) return zero ? launch4<S, UM, true, true>(a, st) : launch4<S, UM, true, false>(a, st);
    return zero ? launch4<S, UM, false, true>(a, st) : launch4<S, UM, false, false>(a, st);
}

template <unsigned UM>
int launch1(TArgs a, int sweeps, bool resid, bool zero, hipStream_t st)
{
    switch (sweeps) {
    case 1: return launch2<1, UM>(a, resid, zero, st);
    case 2: return launch2<2, UM>(a, resid, zero, st);
    default: return launch2<3, UM>(a, resid, zero, st);
    }
}

}  // namespace

int lmg_tile_tune_set(const char *key, int v)
{
    if (strcmp(key, "tile_rows") == 0 || strcmp(key, "tile_rows_big") == 0) {
        if (v != 0 && v != 16 && v != 32) return LMG_ERR_ARG;
        (key[9] ? g_tile_rows_big : g_tile_rows) = v;
        return LMG_OK;
    }
    if (strcmp(key, "tile_prol_wide_lines") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_tile_prol_wide_lines = v;
        return LMG_OK;
    }
    if (strcmp(key, "tile_big_lines") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_tile_big_lines = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_tile_tune_get(const char *key)
{
    if (strcmp(key, "tile_rows") == 0) return g_tile_rows;
    if (strcmp(key, "tile_rows_big") == 0) return g_tile_rows_big;
    if (strcmp(key, "tile_big_lines") == 0) return g_tile_big_lines;
    if (strcmp(key, "tile_prol_wide_lines") == 0) return g_tile_prol_wide_lines;
    return LMG_ERR_ARG;
}

extern "C" {

static int tile_args(TArgs &a, int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                     const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val, int sweeps,
                     const double *x_in, const double *b, double omega, double *x_out, double *r_out)
{
    if (n < 0 || n >= (1ll << 31) - 4096 || npat < 1 || npat > kMaxPat || (union_mask & ~0x1FFu)) return LMG_ERR_ARG;
    if (sweeps < 1 || sweeps > 3) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!pid || !st_val || !st_mask || !b || !x_out || x_in == x_out || r_out == x_out || (r_out && r_out == x_in))
        return LMG_ERR_ARG;
    if (line_stride < 3 || line_stride > n) return LMG_ERR_ARG;
    a.n = (int)n;
    a.W = line_stride;
    a.lines = (int)((n + line_stride - 1) / line_stride);
    a.npat = npat;
    a.tiles_x = a.tiles_y = 0;
    a.pid = pid;
    a.st_val = st_val;
    a.st_mask = st_mask;
    a.x = x_in;
    a.b = b;
    a.out = x_out;
    a.r = r_out;
    a.omega = omega;
    a.hot = -1;
    for (int k = 0; k < 9; ++k) a.hot_val[k] = 0.0;
    a.hot_rdiag = 0.0;
    if (hot_pattern >= 0 && hot_pattern < npat && h_hot_val && h_hot_val[4] != 0.0) {
        a.hot = hot_pattern;
        for (int k = 0; k < 9; ++k) a.hot_val[k] = h_hot_val[k];
        a.hot_rdiag = 1.0 / h_hot_val[4];
    }
    a.ec = nullptr;
    a.nc = a.Wc = 0;
    a.ppid = nullptr;
    a.pp_val = nullptr;
    a.pp_mask = nullptr;
    a.pp_npat = 0;
    a.bc = nullptr;
    a.rpid = nullptr;
    a.rp_val = nullptr;
    a.rp_mask = nullptr;
    a.rp_npat = 0;
    return 1;                                  // filled: launch
}

int lmg_stencil_smooth_tiled(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                             const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val,
                             int sweeps, const double *x_in, const double *b, double omega, double *x_out, double *r_out,
                             void *stream)
{
    TArgs a;
    const int rc = tile_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, r_out);
    if (rc != 1) return rc;
    hipStream_t st = lmg_stream(stream);
    const bool resid = r_out != nullptr, zero = x_in == nullptr;
    switch (union_mask) {
    case kMask5: return launch1<kMask5>(a, sweeps, resid, zero, st);
    case kMask9: return launch1<kMask9>(a, sweeps, resid, zero, st);
    default: return LMG_ERR_CAPACITY;        // other slot sets: run the separate sweeps
    }
}

int lmg_stencil_smooth_tiled_prolong(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                                     const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern,
                                     const double *h_hot_val, int sweeps, const double *x_in, const double *b, double omega,
                                     double *x_out, int64_t n_coarse, int32_t coarse_stride, const double *e_coarse,
                                     const uint8_t *p_pid, int32_t p_npat, const double *p_val, const int32_t *p_mask,
                                     void *stream)
{
    if (!x_in || !e_coarse || !p_pid || !p_val || !p_mask || p_npat < 1 || p_npat > kMaxPat) return LMG_ERR_ARG;
    if (n_coarse < 2 || n_coarse >= (1ll << 31) || coarse_stride < 1 || coarse_stride > n_coarse) return LMG_ERR_ARG;
    if (e_coarse == x_out) return LMG_ERR_ARG;
    TArgs a;
    const int rc = tile_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, nullptr);
    if (rc != 1) return rc;
    a.ec = e_coarse;
    a.nc = (int)n_coarse;
    a.Wc = coarse_stride;
    a.ppid = p_pid;
    a.pp_val = p_val;
    a.pp_mask = p_mask;
    a.pp_npat = p_npat;
    hipStream_t st = lmg_stream(stream);
    switch (union_mask) {
    case kMask5: return launch_prol<kMask5>(a, sweeps, st);
    case kMask9: return launch_prol<kMask9>(a, sweeps, st);
    default: return LMG_ERR_CAPACITY;
    }
}

int lmg_stencil_smooth_tiled_restrict(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                                      const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern,
                                      const double *h_hot_val, int sweeps, const double *x_in, const double *b, double omega,
                                      double *x_out, int64_t n_coarse, int32_t coarse_stride, double *b_coarse,
                                      const uint8_t *r_pid, int32_t r_npat, const double *r_val, const int32_t *r_mask,
                                      void *stream)
{
    if (!b_coarse || !r_pid || !r_val || !r_mask || r_npat < 1 || r_npat > kMaxPat) return LMG_ERR_ARG;
    if (n_coarse < 1 || n_coarse >= (1ll << 31) || coarse_stride < 1 || coarse_stride > n_coarse) return LMG_ERR_ARG;
    if ((const double *)b_coarse == x_in || b_coarse == x_out || (const double *)b_coarse == b) return LMG_ERR_ARG;
    // every fine node (even line, even column) must have its coarse row
    const int64_t lines = n > 0 ? (n + line_stride - 1) / line_stride : 0;
    // -- and nothing else: the pass only writes b_coarse under those nodes, a larger coarse grid would keep stale rows
    if ((int64_t)coarse_stride != ((int64_t)line_stride + 1) / 2 || (n % line_stride) != 0 || n_coarse != ((lines + 1) / 2) * coarse_stride)
        return LMG_ERR_ARG;
    TArgs a;
    const int rc = tile_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, nullptr);
    if (rc != 1) return rc;
    a.bc = b_coarse;
    a.nc = (int)n_coarse;
    a.Wc = coarse_stride;
    a.rpid = r_pid;
    a.rp_val = r_val;
    a.rp_mask = r_mask;
    a.rp_npat = r_npat;
    hipStream_t st = lmg_stream(stream);
    const bool zero = x_in == nullptr;
    switch (union_mask) {
    case kMask5: return launch_rest<kMask5>(a, sweeps, zero, st);
    case kMask9: return launch_rest<kMask9>(a, sweeps, zero, st);
    default: return LMG_ERR_CAPACITY;
    }
}

int lmg_stencil_smooth_tiled_supported(uint32_t union_mask)
{
    return union_mask == kMask5 || union_mask == kMask9;
}

}  // extern "C"
