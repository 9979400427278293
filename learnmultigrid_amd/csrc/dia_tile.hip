// Fused smoothing passes for grid operators with VARIABLE coefficients (gfx950):
//     x_out = J^S(x_in)   [r = b - A x_out]          S = 1..3 weighted-Jacobi sweeps, x_in == NULL: zero iterate
// on the DIA twin of a CSR matrix whose entries all sit at  column - row = c*W + d,  c, d in {-1, 0, 1}  (the 3x3
// slots of stencil.hip) but whose VALUES differ from row to row: variable-coefficient stiffness matrices (cfg#5),
// P1 matrices of jittered triangulations (cfg#3: 7 slots) -- the operators the reference's learned transfers are
// built for (Multigrid.py:306-370, :741-765).  The twin stores one fp64 array of n values per slot of the union
// mask (40 B/row for a 5-point operator instead of 64 B/row of CSR, no indices); a row that lacks a slot holds +0.0
// there, which adds +0.0 * x to its sum -- bitwise neutral for finite x (the sum never is -0.0), exactly like an
// explicit zero in the CSR input.
//
// Same structure as stencil_tile.hip: a WORKGROUP owns a tile of 64 columns x RR lines (it stores the inner
// 64 - 2H x RR - 2H part), every lane owns RB elements of it for ALL sweeps, so the matrix values of its rows are
// loaded ONCE into registers and serve the S sweeps and the residual; only the iterate travels (two LDS buffers).
// One pass over the level reads ~1.5 x (8 slots + 16) B/row instead of (S + 1) x 74 B/row of the packed-CSR sweeps.
// The arithmetic per row and sweep is the sequence of the CSR kernels (entries in ascending column order = slot
// order, separate multiply and add, omega * (rdiag * r)): bit-identical to them and to the CPU oracle.
#include <string.h>
#include "lmg_common.hpp"

namespace {

constexpr int kCols = 64;                     // columns of a tile = lanes of a wave
constexpr int kLS = kCols + 2;                // LDS line stride: one guard column on either side

__device__ __forceinline__ double dpp_lower(double src)      // lane i <- lane i-1, lane 0 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_upper(double src)      // lane i <- lane i+1, lane 63 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct DArgs {
    int n, W, lines;
    int tiles_x, tiles_y;
    const double *dia;        // [nslots][n]
    const double *x;          // may be NULL with ZERO
    const double *b;
    double *out;
    double *r;                // may be NULL without RESID
    double omega;
};

template <unsigned UM> struct Slots {
    static constexpr int count = __builtin_popcount(UM);
    static constexpr int index(int s) { return __builtin_popcount(UM & ((1u << s) - 1u)); }
};

template <int S, unsigned UM, bool RESID, bool ZERO, int RR, int RB>
__global__ void __launch_bounds__(RR / RB * LMG_WAVE) dia_tile_kernel(DArgs a)
{
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0);
    constexpr int kWaves = RR / RB;
    constexpr int NS = Slots<UM>::count;
    constexpr bool DIAG = (UM & 0x145u) != 0;
    static_assert(RR > 2 * H + 1 && kCols > 2 * H && RR % RB == 0 && (UM & 16u), "tile smaller than its halo / no diagonal slot");
    __shared__ double s_x[2][RR * kLS];

    const int t = threadIdx.x, lane = t & (LMG_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tx = (int)blockIdx.x % a.tiles_x, ty = (int)blockIdx.x / a.tiles_x;
    const int c0 = tx * (kCols - 2 * H) - H, y0 = ty * (RR - 2 * H) - H;
    const int n = a.n;
    const int64_t W = a.W;
    const int rb0 = wave * RB;

    // ---- the lane's RB elements: iterate, right-hand side and ALL matrix values of their rows, requested at once ----
    double lx[RB], bk[RB], av[RB][NS];
    bool ok[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int y = y0 + rb0 + k;
        const int64_t i = (int64_t)y * W + c0 + lane;
        ok[k] = y >= 0 && y < a.lines && i >= 0 && i < n;
        const int64_t j = ok[k] ? i : 0;
        lx[k] = (!ZERO && ok[k]) ? a.x[j] : 0.0;
        bk[k] = ok[k] ? a.b[j] : 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) av[k][q] = ok[k] ? a.dia[(int64_t)q * n + j] : 0.0;
    }
    for (int i = t; i < 2 * RR; i += kWaves * LMG_WAVE) {         // guard columns of both iterate buffers
        const int r = i >> 1, g = (i & 1) ? kLS - 1 : 0;
        s_x[0][r * kLS + g] = 0.0;
        s_x[1][r * kLS + g] = 0.0;
    }
    double rd[RB];                                                // 1 / a_ii (0: no usable diagonal -> the sweep copies x)
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        s_x[0][(rb0 + k) * kLS + 1 + lane] = lx[k];
        s_x[1][(rb0 + k) * kLS + 1 + lane] = 0.0;
        const double dg = av[k][Slots<UM>::index(4)];
        rd[k] = dg != 0.0 ? 1.0 / dg : 0.0;
    }
    __syncthreads();

    const double omega = a.omega;
    struct Win { double m, c, p; };
    auto line = [&](const double *src, int r, bool sides) -> Win {
        Win w;
        w.c = src[r * kLS + 1 + lane];
        w.m = sides ? dpp_lower(w.c) : 0.0;
        w.p = sides ? dpp_upper(w.c) : 0.0;
        return w;
    };
    // A x of the centre line of (u, c, d) for element k of this lane, slot order = column order
    auto apply = [&](const Win &u, const Win &c, const Win &d, int k) -> double {
        const double w[9] = {u.m, u.c, u.p, c.m, c.c, c.p, d.m, d.c, d.p};
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < 9; ++s)
            if ((UM >> s) & 1u) acc = acc + av[k][Slots<UM>::index(s)] * w[s];
        return acc;
    };
    auto block = [&](const double *src, auto &&emit) {
        Win ln[RB + 2];
#pragma unroll
        for (int j = 0; j < RB + 2; ++j) {
            const int rr = min(max(rb0 - 1 + j, 0), RR - 1);
            ln[j] = line(src, rr, DIAG || (j >= 1 && j <= RB));
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) emit(k, ln[k + 1].c, apply(ln[k], ln[k + 1], ln[k + 2], k));
    };

    // ---- the sweeps: iterate s goes from buffer (s - 1) & 1 to buffer s & 1 -------------------------------------
#pragma unroll
    for (int s = 1; s <= S; ++s) {
        const double *src = s_x[(s - 1) & 1];
        double *dst = s_x[s & 1];
        if (ZERO && s == 1) {
            // first sweep from a zero iterate: x = omega * (D^-1 b) on every line (lmg_vmul's bits)
#pragma unroll
            for (int k = 0; k < RB; ++k) dst[(rb0 + k) * kLS + 1 + lane] = ok[k] ? omega * (rd[k] * bk[k]) : 0.0;
        } else {
            block(src, [&](int k, double xc, double acc) {
                const double res = bk[k] - acc;
                const double nx = rd[k] == 0.0 ? xc : xc + omega * (rd[k] * res);
                // (lines 0 and RR - 1 have no line above / below: halo, never read where it matters)
                dst[(rb0 + k) * kLS + 1 + lane] = ok[k] ? nx : 0.0;
            });
        }
        __syncthreads();
    }

    // ---- outputs: the inner part of the tile, inside the line, rows of the matrix --------------------------------
    const double *fin = s_x[S & 1];
    const bool col_ok = lane >= H && lane < kCols - H && c0 + lane >= 0 && c0 + lane < W;
    if (RESID) {
        block(fin, [&](int k, double xc, double acc) {
            const int r = rb0 + k;
            const int64_t i = (int64_t)(y0 + r) * W + c0 + lane;
            if (r >= H && r < RR - H && col_ok && ok[k]) {
                a.out[i] = xc;
                a.r[i] = bk[k] - acc;
            }
        });
    } else {
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int r = rb0 + k;
            if (r >= H && r < RR - H && col_ok && ok[k]) a.out[(int64_t)(y0 + r) * W + c0 + lane] = fin[r * kLS + 1 + lane];
        }
    }
}

// DIA twin of a CSR matrix: dia[q][row] = value of the entry in slot number q of the union mask (ascending slots), 0.0
// where the row has none.  One thread per row; an entry that is no 3x3 slot of stride W, or not in the union mask,
// raises the mismatch flag (the caller then keeps the packed CSR).  probe != 0: only OR the slots into mask_out.
__global__ void __launch_bounds__(256) dia_fill_kernel(int64_t n, int W, const int *rowptr, const int *colidx, const double *vals,
                                                       unsigned umask, double *dia, int *mismatch, unsigned *mask_out)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned seen_all = 0;
    bool bad = false;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += stride) {
        double v9[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) v9[s] = 0.0;
        unsigned seen = 0;
        for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
            const int64_t off = (int64_t)colidx[e] - row;
            int slot = -1;
#pragma unroll
            for (int c = -1; c <= 1; ++c) {
                const int64_t d = off - (int64_t)c * W;
                if (d >= -1 && d <= 1 && slot < 0) slot = (c + 1) * 3 + (int)(d + 1);
            }
            if (slot < 0 || ((seen >> slot) & 1u)) {
                bad = true;
                continue;
            }
            seen |= 1u << slot;
            const double v = vals ? vals[e] : 0.0;
#pragma unroll
            for (int s = 0; s < 9; ++s)
                if (s == slot) v9[s] = v;
        }
        seen_all |= seen;
        if (dia) {
            if (seen & ~umask) bad = true;
            int q = 0;
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if ((umask >> s) & 1u) {
                    dia[(int64_t)q * n + row] = v9[s];
                    ++q;
                }
            }
        }
    }
    if (bad) atomicOr(mismatch, 1);
    if (mask_out && seen_all) atomicOr(mask_out, seen_all);
}

int g_dia_rows = 0;             // lines per tile: 0 = default, 32 or 64

template <int S, unsigned UM, bool RESID, bool ZERO, int RR, int RB>
int launch5(DArgs a, hipStream_t st)
{
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0);
    a.tiles_x = (a.W + (kCols - 2 * H) - 1) / (kCols - 2 * H);
    a.tiles_y = (a.lines + (RR - 2 * H) - 1) / (RR - 2 * H);
    const int64_t grid = (int64_t)a.tiles_x * a.tiles_y;
    if (grid > 0x7fffffff) return LMG_ERR_CAPACITY;
    hipLaunchKernelGGL((dia_tile_kernel<S, UM, RESID, ZERO, RR, RB>), dim3((unsigned)grid), dim3(RR / RB * LMG_WAVE), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int S, unsigned UM, bool RESID, bool ZERO>
int launch4(DArgs a, hipStream_t st)
{
    const int rr = g_dia_rows == 0 ? 32 : g_dia_rows;
    if (rr == 64) return launch5<S, UM, RESID, ZERO, 64, 4>(a, st);
    return launch5<S, UM, RESID, ZERO, 32, 2>(a, st);
}

template <int S, unsigned UM>
int launch2(DArgs a, bool resid, bool zero, hipStream_t st)
{
    if (resid) return zero ? launch4<S, UM, true, true>(a, st) : launch4<S, UM, true, false>(a, st);
    return zero ? launch4<S, UM, false, true>(a, st) : launch4<S, UM, false, false>(a, st);
}

template <unsigned UM>
int launch1(DArgs a, int sweeps, bool resid, bool zero, hipStream_t st)
{
    switch (sweeps) {
    case 1: return launch2<1, UM>(a, resid, zero, st);
    case 2: return launch2<2, UM>(a, resid, zero, st);
    default: return launch2<3, UM>(a, resid, zero, st);
    }
}

constexpr unsigned kMask5 = 0x0BAu;           // {-W, -1, 0, +1, +W}
constexpr unsigned kMask7a = 0x1BBu;          // + {-W-1, +W+1}: P1 on triangles cut along one diagonal
constexpr unsigned kMask7b = 0x0FEu;          // + {-W+1, +W-1}: the other diagonal
constexpr unsigned kMask9 = 0x1FFu;

}  // namespace

int lmg_dia_tune_set(const char *key, int v)
{
    if (strcmp(key, "dia_rows") == 0) {
        if (v != 0 && v != 32 && v != 64) return LMG_ERR_ARG;
        g_dia_rows = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_dia_tune_get(const char *key)
{
    if (strcmp(key, "dia_rows") == 0) return g_dia_rows;
    return LMG_ERR_ARG;
}

extern "C" {

int lmg_dia_smooth_supported(uint32_t union_mask)
{
    return union_mask == kMask5 || union_mask == kMask7a || union_mask == kMask7b || union_mask == kMask9;
}

int lmg_dia_fill(int64_t n, int32_t line_stride, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                 uint32_t union_mask, double *dia, int32_t *mismatch, uint32_t *mask_out, void *stream)
{
    if (n < 0 || line_stride < 3 || (union_mask & ~0x1FFu) || (n > 0 && (!rowptr || !colidx || !mismatch))) return LMG_ERR_ARG;
    if (dia && !vals) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(dia_fill_kernel, dim3((unsigned)grid), dim3(256), 0, lmg_stream(stream), n, (int)line_stride, rowptr, colidx,
                       vals, union_mask, dia, mismatch, mask_out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_dia_smooth(int64_t n, int32_t line_stride, uint32_t union_mask, const double *dia, int sweeps, const double *x_in,
                   const double *b, double omega, double *x_out, double *r_out, void *stream)
{
    if (n < 0 || n >= (1ll << 31) - 4096) return LMG_ERR_ARG;
    if (sweeps < 1 || sweeps > 3) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!dia || !b || !x_out || x_in == x_out || r_out == x_out || (r_out && r_out == x_in)) return LMG_ERR_ARG;
    if (line_stride < 3 || line_stride > n) return LMG_ERR_ARG;
    DArgs a;
    a.n = (int)n;
    a.W = line_stride;
    a.lines = (int)((n + line_stride - 1) / line_stride);
    a.tiles_x = a.tiles_y = 0;
    a.dia = dia;
    a.x = x_in;
    a.b = b;
    a.out = x_out;
    a.r = r_out;
    a.omega = omega;
    hipStream_t st = lmg_stream(stream);
    const bool resid = r_out != nullptr, zero = x_in == nullptr;
    switch (union_mask) {
    case kMask5: return launch1<kMask5>(a, sweeps, resid, zero, st);
    case kMask7a: return launch1<kMask7a>(a, sweeps, resid, zero, st);
    case kMask7b: return launch1<kMask7b>(a, sweeps, resid, zero, st);
    case kMask9: return launch1<kMask9>(a, sweeps, resid, zero, st);
    default: return LMG_ERR_CAPACITY;
    }
}

}  // extern "C"
