// Fused smoothing passes on grid-stencil matrices (format of stencil.hip) for gfx950:
//     x_out = J^S(x_in)              S = 1..3 weighted-Jacobi sweeps x <- x + omega * D^-1 (b - A x)
//     r     = b - A x_out            (optional: the residual the cycle restricts next, Multigrid.py:90)
// in ONE pass over x_in, b and the pattern ids -- 25 B/row (+8 with r) for up to four operator
// applications instead of 25 B/row each.  The S sweeps of Multigrid.py:88 / :121 are separate launches of
// stencil_sweep_kernel otherwise; the arithmetic per row and per sweep is the same instruction sequence,
// so every value is bit-identical to the unfused sequence (tests assert array_equal).
//
// Temporal blocking without LDS and without barriers: one WAVE owns a strip of 128 consecutive linear
// indices per grid line (two per lane, 16-byte loads / stores) and marches down a segment of lines with
// all intermediate iterates in registers:
//     step t:  line t of x_in, b, ids arrives (issued PF steps earlier),
//              x^1 line t-1 from x^0 lines t-2..t,  x^2 line t-2 from x^1 lines t-3..t-1, ...,
//              x^S line t-S is stored,  r line t-S-1 from x^S lines t-S-2..t-S is stored.
// Left / right neighbours come from the neighbouring lanes (DPP wave shifts); the window edges are
// simply wrong after each sweep, which is why a strip only STORES its inner 128 - 2H columns and a
// segment its inner lines (H = S, +1 with the residual): strips overlap by 2H columns, segments by 2H
// lines (redundant loads and arithmetic, a few per cent).  Everything is expressed in linear indices
// i = line * W + column, exactly like the format itself: a "column" outside [0, W) is just the linear
// neighbour in the adjacent line, so no grid geometry is assumed beyond the pattern offsets c*W + d.
#include <string.h>
#include "lmg_common.hpp"

#ifndef LMG_FUSED_NT_REST
#define LMG_FUSED_NT_REST 3
#endif
#ifndef LMG_FUSED_NT_PROL
#define LMG_FUSED_NT_PROL 0
#endif
#ifndef LMG_FUSED_NT_PLAIN
#define LMG_FUSED_NT_PLAIN 0
#endif

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / LMG_WAVE;
constexpr int kMaxPat = 64;
constexpr int kStripCols = 2 * LMG_WAVE;      // linear indices per line and wave
constexpr unsigned kMask5 = 0x0BAu;           // slots {-W, -1, 0, +1, +W}
constexpr unsigned kMask9 = 0x1FFu;           // full 3x3
constexpr unsigned kMask1D = 0x038u;          // {-1, 0, +1}

typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) d2u { double a, b; };

struct MArgs {
    int n;
    int W;
    int lines;                // ceil(n / W)
    int npat;
    unsigned umask;
    int strips, segs, seg_lines;
    int items;                // waves of the launch
    int nb_strips;            // boundary strips (first, last) cut into their own, shorter segments and numbered first: 0, 1 or 2
    int segs_b, seg_lines_b;
    int edge_lines;           // length of the first / last segment of the other strips (0: uniform segments)
    int allow_fast;
    const unsigned char *pid;
    const double *st_val;
    const int *st_mask;
    const double *x;          // may be NULL with ZERO
    const double *b;
    double *out;
    double *r;                // may be NULL without RESID
    double omega;
    int hot;                  // most frequent pattern with all union slots and a diagonal, or -1
    double hot_val[9];        // its values by slot (scalar registers in the kernel)
    double hot_rdiag;         // 1 / its diagonal
    // PROL: x_in + P e_c is formed on the fly (the correction of Multigrid.py:115 fused into the post-smoothing pass)
    const double *ec;         // coarse vector
    int nc, Wc;               // its length and line stride: row (y, x) of P reads columns ((y >> 1) * Wc + (x >> 1)) + {0, 1, Wc, Wc + 1}
    const unsigned char *ppid;    // pattern id of every row of P
    const double *pp_val;     // [pp_npat][4] values by slot
    const int *pp_mask;       // [pp_npat] slots present
    int pp_npat;
    int hotp[2];              // even / odd lines: ids (A | B << 8) of the frequent pair of an (even, odd) column pair, or -1
    double hp[9];             // their values: even line A s0 | B s0 s1 || odd line A s0 s2 | B s0 s1 s2 s3
    // REST: b_coarse = R r is formed on the fly instead of storing r (Multigrid.py:90 + :93 fused into the pre-smoothing
    // pass): row (Y, X) of R reads r at lines 2Y - 1 .. 2Y + 1, columns 2X - 1 .. 2X + 1 (slots 0..8 like the operator)
    double *bc;               // coarse right-hand side (output); nc, Wc as above
    const unsigned char *rpid;    // pattern id of every row of R
    const double *rp_val;     // [rp_npat][9]
    const int *rp_mask;       // [rp_npat]
    int rp_npat;
    int hotr;                 // the frequent pattern with all nine slots, or -1
    double hr[9];             // its values
};

// strip geometry: columns left of the stored part / stored columns of a 128-column window
template <int H, bool PROL, bool REST> struct StripGeom {
    // The window starts on an even column and stores an even number of columns: a lane's pair is stored whole or not
    // at all (except for the last column of an odd line stride), and with PROL / REST lane l owns coarse column c0 / 2 + l.
    // PROL: its last pair is never exact (lane 63 has no right-hand coarse neighbour): right margin H + 2 or more.
    // REST: the residual must be exact one line / column beyond the stored part: margins H + 1 or more.
    static constexpr bool EVEN = true;
    static constexpr int MLmin = REST ? H + 1 : H, MRmin = REST ? H + 1 : (PROL ? H + 2 : H);
    static constexpr int ML = EVEN ? ((MLmin + 1) & ~1) : MLmin;
    static constexpr int U = EVEN ? ((kStripCols - ML - MRmin) & ~1) : kStripCols - ML - MRmin;
};

__device__ __forceinline__ double dpp_lower(double src)      // lane i <- lane i-1, lane 0 <- 0
{
    // (bound_ctrl: a lane without a source reads 0 -- no register has to be cleared for it first)
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

__device__ __forceinline__ d2 load2(const double *__restrict__ v, int64_t i, int n)
{
    d2 r;
    if (i >= 0 && i + 1 < n) {
        const d2u t = *reinterpret_cast<const d2u *>(v + i);
        r.x = t.a;
        r.y = t.b;
    } else {
        r.x = (i >= 0 && i < n) ? v[i] : 0.0;
        r.y = (i + 1 >= 0 && i + 1 < n) ? v[i + 1] : 0.0;
    }
    return r;
}

// One line of input as it comes out of memory.  Every step issues EXACTLY the same vector-memory
// instructions (three loads from clamped addresses, two -- with the residual four -- buffer stores whose
// disabled lanes point out of range), none of them inside a branch, and nothing is computed from a load's
// result until the line is consumed PF steps later.  gfx950 counts loads and stores in ONE in-order
// counter (vmcnt): a load or store issued on only one side of a branch makes the compiler wait for the
// smaller of the two counts after the join, i.e. for loads that were meant to stay in flight, and the wave
// serialises on the memory latency once per line (measured: 0.157 -> see DESIGN.md).
// Two later variants lost against this one IN THE CYCLE (same-box A/B of bench.py, cfg#4) although a stand-alone
// timing loop liked them: (a) 32-bit buffer addressing with range-checked loads instead of the clamped 64-bit
// addresses below -- cycle 0.987 vs 0.931 ms (pre-smoothing pass 214 vs 195 us, the 9-point level 100 / 81 vs
// 79 / 63 us: the compiler turns the selects on the loaded halves into branches with their own waits);
// (b) a prologue padded with dropped stores so that the compiler's wait at the loop top stays counted -- no gain.
struct Line {
    d2 x, b;
    double e;       // PROL: coarse line (y + 1) >> 1 at the lane's coarse column
    int pp;         // PROL: the two pattern ids of P as loaded
    int rp;         // REST: the pattern id of R's row (coarse line (y - S) >> 1, the lane's coarse column)
    int praw;       // the two pattern ids as loaded (16 bits)
    int ok;         // bit 0 / 1: element 0 / 1 is a row of the matrix; bit 2 / 3: the pair was clamped
                    // up (i == -1) / down (i == n-1) and holds the wanted element in the other half
};

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOOB = 0xFFFFFFF0u;        // buffer offset beyond any num_records: the access is dropped

// window of one line for the two elements of a lane: [0] = x[. - 1], [1], [2] = the lane's own two, [3] = x[. + 2]
template <bool SIDES>
__device__ __forceinline__ void window(const d2 &c, double (&w)[4])
{
    w[1] = c.x;
    w[2] = c.y;
    if (SIDES) {
        w[0] = dpp_lower(c.y);
        w[3] = dpp_upper(c.x);
    } else {
        w[0] = w[3] = 0.0;
    }
}

// A x for the two elements of a lane on one line, per-lane patterns: slot order = column order
template <unsigned UM>
__device__ __forceinline__ void apply_rows(const double *s_val, int pA, int pB, int mA, int mB,
                                           const d2 &u, const d2 &c, const d2 &d, double &accA, double &accB)
{
    constexpr bool DIAG = (UM & 0x145u) != 0;
    double wu[4], wc[4], wd[4];
    window<true>(c, wc);
    window<DIAG>(u, wu);
    window<DIAG>(d, wd);
    accA = 0.0;
    accB = 0.0;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        if (!((UM >> s) & 1u)) continue;                           // compile time
        const int cc = s / 3, dd = s % 3;
        const double *w = cc == 0 ? wu : (cc == 1 ? wc : wd);
        const double vA = s_val[pA * 9 + s], vB = s_val[pB * 9 + s];
        const double tA = accA + vA * w[dd], tB = accB + vB * w[dd + 1];
        accA = ((mA >> s) & 1) ? tA : accA;
        accB = ((mB >> s) & 1) ? tB : accB;
    }
}

// The same for a line on which every lane holds the HOT pattern (the most frequent one, all union slots
// present): its values sit in scalar registers, no LDS reads, no selects -- the same products and sums
// in the same order, hence the same bits.
template <unsigned UM>
__device__ __forceinline__ void apply_rows_hot(const double (&hv)[9], const d2 &u, const d2 &c,
                                               const d2 &d, double &accA, double &accB)
{
    constexpr bool DIAG = (UM & 0x145u) != 0;
    double wu[4], wc[4], wd[4];
    window<true>(c, wc);
    window<DIAG>(u, wu);
    window<DIAG>(d, wd);
    accA = 0.0;
    accB = 0.0;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        if (!((UM >> s) & 1u)) continue;                           // compile time
        const int cc = s / 3, dd = s % 3;
        const double *w = cc == 0 ? wu : (cc == 1 ? wc : wd);
        accA = accA + hv[s] * w[dd];
        accB = accB + hv[s] * w[dd + 1];
    }
}

#ifdef LMG_FUSED_TRACE
// probe builds (tools/probe/fused_wavetime.hip): cycle counter of the wave of item LMG_FUSED_TRACE at eight points of
// each of its first 64 steps (0 step start, 1 line arrived, 2 next line requested, 3.. after each stage, 7 step end)
__device__ unsigned long long g_fused_trace[64 * 8];
__device__ int g_fused_trace_item;
#define LMG_TRACE(slot)                                                                                   \
    do {                                                                                                  \
        if (FAST && trace_me && (t - y_begin) < 64) g_fused_trace[(t - y_begin) * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define LMG_TRACE(slot)
#endif
#ifdef LMG_FUSED_WAVETIME
__device__ unsigned long long g_fused_wavetime[2 * LMG_FUSED_WAVETIME];   // probe builds (tools/probe/fused_wavetime.hip): start / end (100 MHz) of every wave
#endif

constexpr int kUF = 6;          // steps per loop iteration = period of all register rings
constexpr int kNBR = 6;         // depth of the b / id rings (>= S + 2)

// ---- work decomposition --------------------------------------------------------------------------------------
// One item = one wave = (strip, segment of lines).  Items do not cost the same per line: a wave whose loads all
// stay inside the vectors runs the FAST body (no clamping, no per-element validity, addresses = a scalar row base +
// a per-lane constant) and, on lines where every lane holds the frequent pattern, takes the values from scalar
// registers; the waves of the first / last strip (boundary rows: per-lane patterns from LDS on every line) and of
// the first / last segment (clamped loads) need about twice the instructions per line.  Measured with a per-wave
// clock (tools/probe/fused_wavetime.hip, 4097^2, 3 sweeps + residual): boundary-strip waves took 94 - 164 us where
// the others took 54 - 111, and the launch waited for them (0.181 ms vs 0.140 with no boundary rows at all).  So
// those items get SHORTER segments (about `g_fused_slow_pct` per cent of the steps of a normal item) and are
// numbered first, i.e. start first.
struct Item {
    int strip, out_y0, out_y1;
};

__device__ __forceinline__ Item decode_item(const MArgs &a, int item)
{
    Item it;
    const int nb_items = a.nb_strips * a.segs_b;
    if (item < nb_items) {
        const int sb = item / a.segs_b, k = item - sb * a.segs_b;
        it.strip = sb == 0 ? 0 : a.strips - 1;
        it.out_y0 = k * a.seg_lines_b;
        it.out_y1 = min(a.lines, it.out_y0 + a.seg_lines_b);
        return it;
    }
    const int k = item - nb_items, ns = a.strips - a.nb_strips;
    const int seg = k / ns;
    it.strip = k - seg * ns + (a.nb_strips > 0 ? 1 : 0);
    if (a.edge_lines > 0) {
        if (seg == 0) {
            it.out_y0 = 0;
            it.out_y1 = a.edge_lines;
        } else if (seg == 1) {
            it.out_y0 = a.lines - a.edge_lines;
            it.out_y1 = a.lines;
        } else {
            it.out_y0 = a.edge_lines + (seg - 2) * a.seg_lines;
            it.out_y1 = min(a.lines - a.edge_lines, it.out_y0 + a.seg_lines);
        }
    } else {
        it.out_y0 = seg * a.seg_lines;
        it.out_y1 = min(a.lines, it.out_y0 + a.seg_lines);
    }
    return it;
}

struct Tables {
    const double *val;          // [npat][9]
    const int *mask;            // [npat], bit 16: no usable diagonal
    const double *rdiag;        // [npat]
    const double *pv;           // PROL: [pp_npat][4]
    const int *pm;
    const double *rv;           // REST: [rp_npat][9]
    const int *rm;
};

// The march of one wave down its segment.  FAST: every load of the wave (halo lines, prefetched lines and the coarse
// vectors included) is inside its array and every element is a row of the matrix -- decided once per wave from scalar
// quantities; the general body keeps the clamped, validity-tracking code for the waves at the first / last lines.
// UM: the union slot mask of the matrix, compile time (5-point, 9-point, 1-D chain: anything else runs
// the separate sweeps) -- a run-time mask costs a scalar branch per slot, stage and line.
template <int S, unsigned UM, bool RESID, bool ZERO, int PF, bool PROL, bool REST, bool FAST>
__device__ __forceinline__ void fused_march(const MArgs &a, const Tables &T, const int lane, const int strip, const int out_y0,
                                            const int out_y1)
{
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0);      // halo in lines and columns
    const double *s_val = T.val;
    const int *s_mask = T.mask;
    const double *s_rdiag = T.rdiag;
    const double *s_pv = T.pv;
    const int *s_pm = T.pm;
    const double *s_rv = T.rv;
    const int *s_rm = T.rm;
    constexpr int U = StripGeom<H, PROL, REST>::U, ML = StripGeom<H, PROL, REST>::ML;   // columns a strip stores / its left margin
    constexpr int HL = H + (REST ? 1 : 0);                       // halo in lines
    const int n = a.n;
    const int64_t W = a.W;
    const int c0 = strip * U - ML;                               // linear-index offset of lane 0's first element
    // lines loaded: [y_begin, y_end); the first one is even, so that the parity of a step is a compile-time
    // property of its place in the unrolled block (PROL, REST)
    const int y_begin = out_y0 - HL - ((out_y0 - HL) & 1), y_end = out_y1 + HL;
    const int cidx = 2 * lane;                                   // window column of element 0
    // columns this lane may store (element 0 / 1): inside the strip's inner part and inside the line
    const bool colA = cidx >= ML && cidx < ML + U && c0 + cidx < W;
    const bool colB = cidx + 1 >= ML && cidx + 1 < ML + U && c0 + cidx + 1 < W;
    const double omega = a.omega;
    // hot pattern: id on both elements + both marked as rows; -1 (no hot pattern) never matches
    const int hot2 = a.hot >= 0 ? (FAST ? (a.hot | (a.hot << 8)) : (a.hot | (a.hot << 8) | (3 << 16))) : -1;
    double hv[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) hv[s] = a.hot_val[s];
    const double hrd = a.hot_rdiag;
    double hp[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) hp[s] = PROL ? a.hp[s] : 0.0;
    const int okbits = FAST ? 0 : (3 << 16);
    const int hotq_even = (PROL && a.hotp[0] >= 0) ? (a.hotp[0] | okbits) : -1;
    const int hotq_odd = (PROL && a.hotp[1] >= 0) ? (a.hotp[1] | okbits) : -1;
    double e_prev = 0.0;                                          // PROL: the coarse line the previous (even) line brought

    // FAST: byte offsets of the lane inside a line's window; lanes that never store point out of range
    const unsigned lane16 = (unsigned)lane * 16u, lane8 = (unsigned)lane * 8u, lane2 = (unsigned)lane * 2u;
    const unsigned st_off_ab = (colA && colB) ? lane16 : kOOB;    // both elements: one 16-byte store
    const unsigned st_off_a = (colA && !colB) ? lane16 : kOOB;    // element 0 only (the last column of an odd line stride)
    const unsigned st_off_c = colA ? lane8 : kOOB;                // REST: the coarse row under element 0

    // Nontemporal loads of the streamed vectors in the FAST body, per variant (bit 0: b, bit 1: x, bit 2: pattern ids); measured
    // on one box at 4097^2: restricting pass 0.141 - 0.144 ms without, 0.137 with b, 0.119 with b and x; the correcting and the
    // plain pass lose 4 - 8 % with the same hints (their halo re-reads come from the caches the hint bypasses)
    constexpr int NTL = REST ? LMG_FUSED_NT_REST : (PROL ? LMG_FUSED_NT_PROL : LMG_FUSED_NT_PLAIN);
    auto fetch = [&](int y, Line &L) {
        if (FAST) {
            // all of it in range: scalar row base + lane offset, nothing to clamp or to mark
#ifdef LMG_FUSED_HALOPROBE
            // TIMING PROBE (wrong results): halo lines are read from the segment's own first / last line -- the same
            // instructions, but no line is fetched by two waves
            const int64_t i0 = (int64_t)min(max(y, out_y0), out_y1 - 1) * W + c0;
#else
            const int64_t i0 = (int64_t)y * W + c0;
#endif
            L.ok = 3;
            const unsigned short *pp2 = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(a.pid + i0) + lane2);
            L.praw = (int)((NTL & 4) ? __builtin_nontemporal_load(pp2) : *pp2);
            typedef double dv2 __attribute__((ext_vector_type(2), aligned(8)));    // (an odd line stride: 8-byte aligned windows)
            const dv2 *pb = reinterpret_cast<const dv2 *>(reinterpret_cast<const char *>(a.b + i0) + lane16);
            const dv2 bb = (NTL & 1) ? __builtin_nontemporal_load(pb) : *pb;
            L.b.x = bb.x;
            L.b.y = bb.y;
            if (!ZERO) {
                const dv2 *px = reinterpret_cast<const dv2 *>(reinterpret_cast<const char *>(a.x + i0) + lane16);
                const dv2 xx = (NTL & 2) ? __builtin_nontemporal_load(px) : *px;
                L.x.x = xx.x;
                L.x.y = xx.y;
            } else {
                L.x.x = L.x.y = 0.0;
            }
            if (PROL) {
                unsigned short twop;
                __builtin_memcpy(&twop, reinterpret_cast<const char *>(a.ppid + i0) + lane2, 2);
                L.pp = (int)twop;
                const int64_t jc0 = (int64_t)((y + 1) >> 1) * a.Wc + (c0 >> 1);
                L.e = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.ec + jc0) + lane8);
            } else {
                L.pp = 0;
                L.e = 0.0;
            }
            if (REST) {
                const int64_t jr0 = (int64_t)((y - S) >> 1) * a.Wc + (c0 >> 1);
                L.rp = (int)*(reinterpret_cast<const unsigned char *>(a.rpid + jr0) + lane);
            } else {
                L.rp = 0;
            }
            return;
        }
        const bool line_ok = y >= 0 && y < a.lines && y < y_end;             // wave-uniform, no branch on it
        const int yc = min(max(y, 0), a.lines - 1);
        const int64_t i = (int64_t)yc * W + c0 + cidx;
        const bool okA = line_ok && i >= 0 && i < n, okB = line_ok && i + 1 >= 0 && i + 1 < n;
        const int64_t j = min(max(i, (int64_t)0), (int64_t)n - 2);           // always a valid pair (n >= 2)
        L.ok = (okA ? 1 : 0) | (okB ? 2 : 0) | (i == -1 ? 4 : 0) | (i == (int64_t)n - 1 ? 8 : 0);
        unsigned short two;
        __builtin_memcpy(&two, a.pid + j, 2);
        L.praw = (int)two;
        const d2u bb = *reinterpret_cast<const d2u *>(a.b + j);
        L.b.x = bb.a;
        L.b.y = bb.b;
        if (!ZERO) {
            const d2u xx = *reinterpret_cast<const d2u *>(a.x + j);
            L.x.x = xx.a;
            L.x.y = xx.b;
        } else {
            L.x.x = L.x.y = 0.0;
        }
        if (PROL) {
            unsigned short twop;
            __builtin_memcpy(&twop, a.ppid + j, 2);
            L.pp = (int)twop;
            // every line brings ONE coarse line, (y + 1) >> 1: its own for an even y, the one below for an odd y
            // (whose upper one came with line y - 1); clamped, like everything else, where nothing is there
            const int64_t jc = (int64_t)((yc + 1) >> 1) * a.Wc + (c0 >> 1) + lane;
            L.e = a.ec[min(max(jc, (int64_t)0), (int64_t)a.nc - 1)];
        } else {
            L.pp = 0;
            L.e = 0.0;
        }
        if (REST) {
            // the step that receives line y computes the residual of line y - S - 1 and, when that one is odd, starts
            // the coarse line (y - S) / 2 below it: its ids travel with line y
            const int64_t jr = (int64_t)((y - S) >> 1) * a.Wc + (c0 >> 1) + lane;
            L.rp = (int)a.rpid[min(max(jr, (int64_t)0), (int64_t)a.nc - 1)];
        } else {
            L.rp = 0;
        }
    };
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)n * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((RESID && !REST) ? a.r : a.out, 0, (int)((unsigned)n * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bc = __builtin_amdgcn_make_buffer_rsrc(REST ? a.bc : a.out, 0, (int)((unsigned)(REST ? a.nc : n) * 8u), 0x00020000);
    double hr[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) hr[s] = REST ? a.hr[s] : 0.0;
    double acc_cur = 0.0;                                         // REST: running sum of the coarse row in progress,
    int rp_cur = 0, rp_now = 0;                                   //       its pattern id / the id that came with this line,
    bool st_cur = false, hot_cur = false;                         //       whether this lane stores it / all storing lanes are hot
    bool sty_cur = false;                                         //       (FAST) whether its line is one this wave stores
    auto store2 = [&](const __amdgpu_buffer_rsrc_t &rs, double *base, int y, int p2, double va, double vb) {
        const bool yok = y >= out_y0 && y < out_y1;
        u4 v4;
        v4.x = (unsigned)__double2loint(va);
        v4.y = (unsigned)__double2hiint(va);
        v4.z = (unsigned)__double2loint(vb);
        v4.w = (unsigned)__double2hiint(vb);
        if (FAST) {
            // a descriptor of the line's window (scalar work); a line that is not stored gets an empty one
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(base + ((int64_t)y * W + c0), 0, yok ? 8 * kStripCols : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v4, rl, st_off_ab, 0, 0);
            u2 v2;
            v2.x = v4.x;
            v2.y = v4.y;
            __builtin_amdgcn_raw_buffer_store_b64(v2, rl, st_off_a, 0, 0);
            return;
        }
        const int64_t i = (int64_t)y * W + c0 + cidx;
        const bool stA = yok && colA && ((p2 >> 16) & 1), stB = yok && colB && ((p2 >> 17) & 1);
        __builtin_amdgcn_raw_buffer_store_b128(v4, rs, (stA && stB) ? (unsigned)i * 8u : kOOB, 0, 0);
        u2 v2;
        v2.x = stA ? v4.x : v4.z;
        v2.y = stA ? v4.y : v4.w;
        __builtin_amdgcn_raw_buffer_store_b64(v2, rs, (stA != stB) ? (unsigned)(stA ? i : i + 1) * 8u : kOOB, 0, 0);
    };

    // Register rings, all indexed with compile-time slots inside the unrolled block of kUF steps:
    // line L of iterate s lives in X[s][L mod 3], its b / ids / "all lanes hot" flag in slot L mod kNBR
    // (phases relative to the first step of a block, which is a multiple of kUF lines after y_begin).
    d2 X[S + 1][3];
    d2 Bq[kNBR];
    int Pq[kNBR];
    bool Hq[kNBR];
    const d2 zero2 = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s <= S; ++s) X[s][0] = X[s][1] = X[s][2] = zero2;
#pragma unroll
    for (int k = 0; k < kNBR; ++k) {
        Bq[k] = zero2;
        Pq[k] = 0;
        Hq[k] = false;
    }

    Line pre[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(y_begin + u, pre[u]);

    const int t_last = out_y1 - 1 + S + (RESID ? 1 : 0) + (REST ? 1 : 0);   // last step that still produces output
#ifdef LMG_FUSED_TRACE
    const bool trace_me = lane == 0 && strip == g_fused_trace_item / 100000 && out_y0 <= g_fused_trace_item % 100000 &&
                          g_fused_trace_item % 100000 < out_y1;            // (strip * 100000 + a line of the segment)
#endif
    for (int tb = y_begin; tb <= t_last; tb += kUF) {
#pragma unroll
        for (int u = 0; u < kUF; ++u) {
            const int t = tb + u;
            LMG_TRACE(0);
            // ---- line t arrives ------------------------------------------------------------------
            const int q0 = u % kNBR;
            {
                const Line &L = pre[u % PF];
                const bool okA = FAST || (L.ok & 1), okB = FAST || (L.ok & 2), up = !FAST && (L.ok & 4), down = !FAST && (L.ok & 8);
                d2 bv, xv;
                int pa, pb;
                if (FAST) {
                    bv = L.b;
                    xv = L.x;
                    pa = L.praw & 0xff;
                    pb = L.praw >> 8;
                } else {
                    // (selects that only do something in the first / last lines of the matrix)
                    bv.x = okA ? (down ? L.b.y : L.b.x) : 0.0;
                    bv.y = okB ? (up ? L.b.x : L.b.y) : 0.0;
                    xv.x = okA ? (down ? L.x.y : L.x.x) : 0.0;
                    xv.y = okB ? (up ? L.x.x : L.x.y) : 0.0;
                    pa = down ? (L.praw >> 8) & 0xff : L.praw & 0xff;
                    pb = up ? L.praw & 0xff : (L.praw >> 8) & 0xff;
                }
                if (REST) rp_now = L.rp;
                if (PROL) {
                    // x + P e_c for the two elements: row (y, x) of P reads e_c at ((y >> 1), (x >> 1)) + {0, 1} x {0, 1};
                    // both elements of a lane share x >> 1 = c0 / 2 + lane.  Same sums in the same order as
                    // lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 1): acc = 0; acc = acc + v * e per entry; x + acc.
                    const bool odd = (u & 1) != 0;                        // y_begin is even: compile time after unrolling
                    int qa, qb, q2;
                    if (FAST) {
                        qa = L.pp & 0xff;
                        qb = L.pp >> 8;
                        q2 = L.pp;
                    } else {
                        qa = down ? (L.pp >> 8) & 0xff : L.pp & 0xff;
                        qb = up ? L.pp & 0xff : (L.pp >> 8) & 0xff;
                        q2 = (okA ? qa : 0) | ((okB ? qb : 0) << 8) | ((L.ok & 3) << 16);
                    }
                    const double en = L.e;
                    const double e0 = odd ? e_prev : en, e1 = en;
                    const double e0r = dpp_upper(e0), e1r = dpp_upper(e1);
                    double accA, accB;
                    if (__all(q2 == (odd ? hotq_odd : hotq_even))) {          // wave-uniform
                        if (!odd) {
                            accA = 0.0 + hp[0] * e0;
                            accB = (0.0 + hp[1] * e0) + hp[2] * e0r;
                        } else {
                            accA = (0.0 + hp[3] * e0) + hp[4] * e1;
                            accB = (((0.0 + hp[5] * e0) + hp[6] * e0r) + hp[7] * e1) + hp[8] * e1r;
                        }
                    } else {
                        const int mA = s_pm[qa], mB = s_pm[qb];
                        const double w[4] = {e0, e0r, e1, e1r};
                        accA = 0.0;
                        accB = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const double tA = accA + s_pv[qa * 4 + k] * w[k], tB = accB + s_pv[qb * 4 + k] * w[k];
                            accA = ((mA >> k) & 1) ? tA : accA;
                            accB = ((mB >> k) & 1) ? tB : accB;
                        }
                    }
                    xv.x = okA ? xv.x + accA : 0.0;
                    xv.y = okB ? xv.y + accB : 0.0;
                    e_prev = en;
                }
                Bq[q0] = bv;
                X[0][u % 3] = xv;
                Pq[q0] = FAST ? L.praw : ((okA ? pa : 0) | ((okB ? pb : 0) << 8) | ((L.ok & 3) << 16));
                Hq[q0] = __all(Pq[q0] == hot2);
            }
            LMG_TRACE(1);
            fetch(t + PF, pre[u % PF]);
            LMG_TRACE(2);
            // ---- stages 1..S: iterate s on line t - s -----------------------------------------------
#pragma unroll
            for (int s = 1; s <= S; ++s) {
                const int q = ((u - s) % kNBR + kNBR) % kNBR;                 // ring slot of line t - s
                const int xm = ((u - s - 1) % 3 + 3) % 3, xc = ((u - s) % 3 + 3) % 3, xp = ((u - s + 1) % 3 + 3) % 3;
                const int p2 = Pq[q];
                const d2 bq = Bq[q];
                d2 nx;
                if (Hq[q]) {                                                 // wave-uniform
                    if (ZERO && s == 1) {
                        nx.x = omega * (hrd * bq.x);
                        nx.y = omega * (hrd * bq.y);
                    } else {
                        double accA, accB;
                        apply_rows_hot<UM>(hv, X[s - 1][xm], X[s - 1][xc], X[s - 1][xp], accA, accB);
                        nx.x = X[s - 1][xc].x + omega * (hrd * (bq.x - accA));
                        nx.y = X[s - 1][xc].y + omega * (hrd * (bq.y - accB));
                    }
                } else {
                    const int pA = p2 & 0xff, pB = (p2 >> 8) & 0xff;
                    const int mA = s_mask[pA], mB = s_mask[pB];
                    if (ZERO && s == 1) {
                        // first sweep from a zero iterate: x = omega * (D^-1 b)  (lmg_vmul's bits)
                        nx.x = omega * (s_rdiag[pA] * bq.x);
                        nx.y = omega * (s_rdiag[pB] * bq.y);
                    } else {
                        double accA, accB;
                        apply_rows<UM>(s_val, pA, pB, mA, mB, X[s - 1][xm], X[s - 1][xc], X[s - 1][xp], accA, accB);
                        const double xa = X[s - 1][xc].x, xb = X[s - 1][xc].y;
                        const double rA = bq.x - accA, rB = bq.y - accB;
                        nx.x = (mA >> 16) ? xa : xa + omega * (s_rdiag[pA] * rA);
                        nx.y = (mB >> 16) ? xb : xb + omega * (s_rdiag[pB] * rB);
                    }
                    if (!FAST) {
                        // elements that are no rows of the matrix stay zero (nothing valid ever reads them)
                        if (!((p2 >> 16) & 1)) nx.x = 0.0;
                        if (!((p2 >> 17) & 1)) nx.y = 0.0;
                    }
                }
                X[s][xc] = nx;
                if (s == S) store2(rs_out, a.out, t - S, p2, nx.x, nx.y);
                LMG_TRACE(2 + s);
            }
            // ---- residual of the final iterate on line t - S - 1 ---------------------------------------
            if (RESID) {
                const int y = t - S - 1;
                const int q = ((u - S - 1) % kNBR + kNBR) % kNBR;
                const int xm = ((u - S - 2) % 3 + 3) % 3, xc = ((u - S - 1) % 3 + 3) % 3, xp = ((u - S) % 3 + 3) % 3;
                const int p2 = Pq[q];
                double accA, accB;
                if (Hq[q]) {
                    apply_rows_hot<UM>(hv, X[S][xm], X[S][xc], X[S][xp], accA, accB);
                } else {
                    const int pA = p2 & 0xff, pB = (p2 >> 8) & 0xff;
                    apply_rows<UM>(s_val, pA, pB, s_mask[pA], s_mask[pB], X[S][xm], X[S][xc], X[S][xp], accA, accB);
                }
                if (!REST) {
                    store2(rs_r, a.r, y, p2, Bq[q].x - accA, Bq[q].y - accB);
                } else {
                    // b_c = R r without storing r.  Row (Y, X) of R sums its entries in column order: slots 0..2 on line
                    // 2Y - 1, 3..5 on line 2Y, 6..8 on line 2Y + 1 -- so the sum simply continues as the residual lines
                    // go by: an ODD line finishes row (y - 1) / 2 (and stores it) and starts row (y + 1) / 2, an even
                    // line adds the middle of its own row.  Two running sums per lane, no ring of residual lines; the
                    // same products and sums in the same order as lmg_rpat_sweep_grid(SPMV, alpha = 1, beta = 0).
                    const double rA = Bq[q].x - accA, rB = Bq[q].y - accB;
                    const double wl[3] = {dpp_lower(rB), rA, rB};              // columns 2X - 1, 2X, 2X + 1
                    const bool y_odd = ((u - S - 1) & 1) != 0;                 // y_begin is even: compile time
                    unsigned off_bc = kOOB;
                    bool emit_line = false;                                    // FAST: the finished row's line is stored by this wave
                    double emit = 0.0;
                    if (!y_odd) {
                        if (hot_cur) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) acc_cur = acc_cur + hr[3 + k] * wl[k];
                        } else {
                            const int mk = s_rm[rp_cur];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const double tv = acc_cur + s_rv[rp_cur * 9 + 3 + k] * wl[k];
                                acc_cur = ((mk >> (3 + k)) & 1) ? tv : acc_cur;
                            }
                        }
                    } else {
                        // finish the row above ...
                        if (hot_cur) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) acc_cur = acc_cur + hr[6 + k] * wl[k];
                        } else {
                            const int mk = s_rm[rp_cur];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const double tv = acc_cur + s_rv[rp_cur * 9 + 6 + k] * wl[k];
                                acc_cur = ((mk >> (6 + k)) & 1) ? tv : acc_cur;
                            }
                        }
                        emit = acc_cur;
                        emit_line = sty_cur;
                        if (!FAST && st_cur) off_bc = (unsigned)(((y - 1) >> 1) * a.Wc + (c0 >> 1) + lane) * 8u;
                        // ... and start the row below: this wave owns it if it owns its fine node (y + 1, c0 + cidx)
                        rp_cur = rp_now;
                        sty_cur = y + 1 >= out_y0 && y + 1 < out_y1;
                        st_cur = sty_cur && colA;
                        hot_cur = __all(!st_cur || rp_cur == a.hotr);
                        acc_cur = 0.0;
                        if (hot_cur) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) acc_cur = acc_cur + hr[k] * wl[k];
                        } else {
                            const int mk = s_rm[rp_cur];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const double tv = acc_cur + s_rv[rp_cur * 9 + k] * wl[k];
                                acc_cur = ((mk >> k) & 1) ? tv : acc_cur;
                            }
                        }
                    }
                    u2 v2;
                    v2.x = (unsigned)__double2loint(emit);
                    v2.y = (unsigned)__double2hiint(emit);
                    if (FAST) {
                        // (an even line stores nothing: empty descriptor, like a row of a line another wave owns)
                        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
                            a.bc + ((int64_t)((y - 1) >> 1) * a.Wc + (c0 >> 1)), 0, (y_odd && emit_line) ? 8 * LMG_WAVE : 0, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b64(v2, rl, st_off_c, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(v2, rs_bc, off_bc, 0, 0);
                    }
                }
            }
            LMG_TRACE(7);
        }
    }
}

template <int S, unsigned UM, bool RESID, bool ZERO, int PF, bool PROL = false, bool REST = false>
// (the restricting pass of three sweeps needs 172 VGPRs: capped at 168 it spilled an address of the general body and lost
// scheduling freedom everywhere -- measured on one box 0.149 ms at 3 waves / SIMD against 0.138 at 2 with 2 560 waves)
__global__ void __launch_bounds__(kBlock, (REST && S >= 3) ? 2 : 3) stencil_fused_kernel(MArgs a)
{
    static_assert(!PROL || (!RESID && !ZERO), "the correction is fused into post-smoothing passes only");
    static_assert(!REST || (RESID && !PROL), "the restriction replaces the store of the residual");
    static_assert(kUF % 3 == 0 && kUF % kNBR == 0 && kUF % PF == 0 && S + 2 <= kNBR, "ring periods");
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0);      // halo in lines and columns
    __shared__ double s_val[kMaxPat * 9];
    __shared__ int s_mask[kMaxPat];
    __shared__ double s_rdiag[kMaxPat];
    __shared__ double s_pv[PROL ? kMaxPat * 4 : 1];
    __shared__ int s_pm[PROL ? kMaxPat : 1];
    __shared__ double s_rv[REST ? kMaxPat * 9 : 1];
    __shared__ int s_rm[REST ? kMaxPat : 1];

    const int t_ = threadIdx.x;
    for (int i = t_; i < a.npat * 9; i += kBlock) s_val[i] = a.st_val[i];
    for (int i = t_; i < a.npat; i += kBlock) {
        const int m = a.st_mask[i];
        const double dg = (m & 16) ? a.st_val[i * 9 + 4] : 0.0;
        s_rdiag[i] = dg != 0.0 ? 1.0 / dg : 0.0;
        s_mask[i] = dg == 0.0 ? (m | (1 << 16)) : m;             // bit 16: no usable diagonal -> copy x
    }
    if (PROL) {
        for (int i = t_; i < a.pp_npat * 4; i += kBlock) s_pv[i] = a.pp_val[i];
        for (int i = t_; i < a.pp_npat; i += kBlock) s_pm[i] = a.pp_mask[i];
    }
    if (REST) {
        for (int i = t_; i < a.rp_npat * 9; i += kBlock) s_rv[i] = a.rp_val[i];
        for (int i = t_; i < a.rp_npat; i += kBlock) s_rm[i] = a.rp_mask[i];
    }
    __syncthreads();

    const int lane = t_ & (LMG_WAVE - 1);
    // the wave index is the same in all lanes: say so, and everything derived from it (strip, segment, lines, row
    // bases, loop bounds) lives in scalar registers and costs no vector instruction
    const int item = (int)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(t_ >> 6);
    if (item >= a.items) return;
    const Item it = decode_item(a, item);
    const Tables T = {s_val, s_mask, s_rdiag, s_pv, s_pm, s_rv, s_rm};

    // FAST?  Every line the march touches -- [y_begin, last prefetched line] -- and the coarse lines that travel
    // with them must lie strictly inside the vectors, for all 128 window columns of the strip.
    constexpr int U = StripGeom<H, PROL, REST>::U, ML = StripGeom<H, PROL, REST>::ML;
    constexpr int HL = H + (REST ? 1 : 0);
    const int c0 = it.strip * U - ML;
    const int y_begin = it.out_y0 - HL - ((it.out_y0 - HL) & 1);
    const int t_last = it.out_y1 - 1 + S + (RESID ? 1 : 0) + (REST ? 1 : 0);
    const int y_fetch = y_begin + ((t_last - y_begin) / kUF + 1) * kUF - 1 + PF;     // last line fetched
    bool fast = a.allow_fast && y_begin >= (REST ? S + 2 : 1) && y_fetch <= a.lines - 2 &&
                (int64_t)y_begin * a.W + c0 >= 0 && (int64_t)y_fetch * a.W + c0 + kStripCols < (int64_t)a.n;
    if (PROL) fast = fast && (int64_t)((y_fetch + 1) >> 1) * a.Wc + (c0 >> 1) + LMG_WAVE <= (int64_t)a.nc;
    if (REST) fast = fast && (int64_t)((y_fetch - S) >> 1) * a.Wc + (c0 >> 1) + LMG_WAVE <= (int64_t)a.nc;
#ifdef LMG_FUSED_WAVETIME
    const unsigned long long wt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (fast) fused_march<S, UM, RESID, ZERO, PF, PROL, REST, true>(a, T, lane, it.strip, it.out_y0, it.out_y1);
    else fused_march<S, UM, RESID, ZERO, PF, PROL, REST, false>(a, T, lane, it.strip, it.out_y0, it.out_y1);
#ifdef LMG_FUSED_WAVETIME
    if (lane == 0 && item < LMG_FUSED_WAVETIME) {
        g_fused_wavetime[2 * item] = wt0;
        g_fused_wavetime[2 * item + 1] = (__builtin_amdgcn_s_memrealtime() & ~3ull) | (fast ? 1ull : 0ull) | 2ull;
    }
#endif
}

int g_fused_seg_lines = 0;      // 0 = chosen per launch
int g_fused_seg_min_lines = 0;  // fused_seg_lines applies to grids of this many lines (tuning one level of a cycle)
int g_fused_seg_max_lines = 0x7fffffff;
int g_fused_pf = 0;             // 0 = default
int g_fused_seg_lines_prol = 0; // segment length of the passes with the correction / the restriction folded in (0 = like the others)
int g_fused_seg_lines_rest = 0;
int g_fused_want_waves = 5120;  // waves a launch aims at when it cuts the lines into segments ...
int g_fused_want_waves_rest3 = 2700;   // ... the passes of three sweeps with a transfer folded in (restricting: 2 waves / SIMD); in the cfg#4 cycle 52 - 54-line
                                       // segments measure 0.486 ms, 44 - 50 and 56 - 58 lines 0.489 - 0.495, 28 lines 0.511
int g_fused_floor_halos = 4;    // ... which are never shorter than this many halos (the redundant lines of a segment: 2 H)
int g_fused_balance = 1;        // shorter segments for the items that run the slower bodies (boundary strips, first / last segment)
int g_fused_slow_pct = 55;      // their steps, per cent of a normal item's
int g_fused_fast = 1;           // 0: every wave runs the general body (tests)

template <int S, unsigned UM, bool RESID, bool ZERO, bool PROL = false, bool REST = false>
int launch4(MArgs a, hipStream_t st)
{
    constexpr int H = S + (RESID ? 1 : 0) - (ZERO ? 1 : 0);
    constexpr int U = StripGeom<H, PROL, REST>::U;
    constexpr int HL = H + (REST ? 1 : 0);
    a.strips = (a.W + U - 1) / U;
    // Segments: about 1.25 x the waves the chip holds at once (256 CUs x 16), never shorter than g_fused_floor_halos
    // halos.  Scanned in the cycle on one box (tools/ab_cycle.py, cfg#4): at 4097^2 the cycle takes 0.885 ms with
    // 28-line segments (5 145 waves), 0.90 with 24 or 47, 0.91 - 0.95 with 32 - 42; at 2049^2 the floor (12 lines) is
    // best, 0.91 vs 0.93 with 10 or 14 - 16 lines.
    int seg_lines = (a.lines >= g_fused_seg_min_lines && a.lines <= g_fused_seg_max_lines) ? g_fused_seg_lines : 0;
    if (PROL && g_fused_seg_lines_prol > 0) seg_lines = g_fused_seg_lines_prol;
    if (REST && g_fused_seg_lines_rest > 0) seg_lines = g_fused_seg_lines_rest;
    if (seg_lines <= 0) {
        // (round 2 gave the pass with the restriction folded in one round of waves with 48-line segments; with the
        // balanced decomposition 28 lines measure best for it too: cycle 0.500 vs 0.509 ms)
        // (the correcting pass too: in the cfg#4 cycle 53-line segments 0.4796 ms, 28 lines 0.486, 20 - 56 lines 0.482 - 0.494)
        const int want = ((REST || PROL) && S >= 3 && !ZERO) ? g_fused_want_waves_rest3 : g_fused_want_waves;
        const int want_segs = (want + a.strips - 1) / a.strips;
        seg_lines = (a.lines + want_segs - 1) / want_segs;
        const int floor_lines = H > 0 ? g_fused_floor_halos * H : 4;
        if (seg_lines < floor_lines) seg_lines = floor_lines;
    }
    a.seg_lines = seg_lines;
    a.allow_fast = g_fused_fast;
    a.nb_strips = 0;
    a.segs_b = 1;
    a.seg_lines_b = seg_lines;
    a.edge_lines = 0;
    if (g_fused_balance) {
        const int steps = seg_lines + 2 * HL;
        int slow = (steps * g_fused_slow_pct) / 100 - 2 * HL;
        if (slow < 2) slow = 2;
        if (slow < seg_lines) {
            a.nb_strips = a.strips < 2 ? a.strips : 2;
            a.seg_lines_b = slow;
            a.segs_b = (a.lines + slow - 1) / slow;
            // the first and the last segment of the other strips: long enough for their neighbours' prefetches to stay
            // inside the grid, i.e. for those to run the FAST body
            int edge = slow < HL + 2 + kUF + 2 ? HL + 2 + kUF + 2 : slow;
            if (a.strips > a.nb_strips && a.lines >= 2 * edge + seg_lines) a.edge_lines = edge;
        }
    }
    if (a.edge_lines > 0) a.segs = 2 + (a.lines - 2 * a.edge_lines + seg_lines - 1) / seg_lines;
    else a.segs = (a.lines + seg_lines - 1) / seg_lines;
    a.items = a.nb_strips * a.segs_b + (a.strips - a.nb_strips) * a.segs;
    const int grid = (a.items + kWavesPerBlock - 1) / kWavesPerBlock;
    // (deeper prefetch, measured round 3 on the scalarised kernel, 4097^2: 3 lines spills in the restricting pass, 0.238 ms;
    // 6 lines at 2 waves / SIMD with 48 - 64-line segments 0.143 vs 0.149 ms with 2 lines and 28-line segments, the
    // correcting pass unchanged -- a wave's step is bound by the dependent chain of its S + 1 stages, not by the loads)
    hipLaunchKernelGGL((stencil_fused_kernel<S, UM, RESID, ZERO, 2, PROL, REST>), dim3((unsigned)grid), dim3(kBlock), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int S, unsigned UM>
int launch2(MArgs a, bool resid, bool zero, hipStream_t st)
{
    if (resid) return zero ? launch4<S, UM, true, true>(a, st) : launch4<S, UM, true, false>(a, st);
    return zero ? launch4<S, UM, false, true>(a, st) : launch4<S, UM, false, false>(a, st);
}

template <unsigned UM>
int launch_prol(MArgs a, int sweeps, hipStream_t st)
{
    switch (sweeps) {
    case 1: return launch4<1, UM, false, false, true>(a, st);
    case 2: return launch4<2, UM, false, false, true>(a, st);
    default: return launch4<3, UM, false, false, true>(a, st);
    }
}

template <unsigned UM>
int launch_rest(MArgs a, int sweeps, bool zero, hipStream_t st)
{
    switch (sweeps) {
    case 1: return zero ? launch4<1, UM, true, true, false, true>(a, st) : launch4<1, UM, true, false, false, true>(a, st);
    case 2: return zero ? launch4<2, UM, true, true, false, true>(a, st) : launch4<2, UM, true, false, false, true>(a, st);
    default: return zero ? launch4<3, UM, true, true, false, true>(a, st) : launch4<3, UM, true, false, false, true>(a, st);
    }
}

template <unsigned UM>
int launch1(MArgs a, int sweeps, bool resid, bool zero, hipStream_t st)
{
    switch (sweeps) {
    case 1: return launch2<1, UM>(a, resid, zero, st);
    case 2: return launch2<2, UM>(a, resid, zero, st);
    default: return launch2<3, UM>(a, resid, zero, st);
    }
}

}  // namespace

int lmg_fused_tune_set(const char *key, int v)
{
    if (strcmp(key, "fused_seg_lines") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_fused_seg_lines = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_pf") == 0) {
        if (v != 0 && v != 2) return LMG_ERR_ARG;
        g_fused_pf = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_seg_min_lines") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_fused_seg_min_lines = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_seg_lines_prol") == 0 || strcmp(key, "fused_seg_lines_rest") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        (key[16] == 'p' ? g_fused_seg_lines_prol : g_fused_seg_lines_rest) = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_seg_max_lines") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_fused_seg_max_lines = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_want_waves_rest3") == 0) {
        if (v < 1) return LMG_ERR_ARG;
        g_fused_want_waves_rest3 = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_want_waves") == 0) {
        if (v < 1) return LMG_ERR_ARG;
        g_fused_want_waves = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_floor_halos") == 0) {
        if (v < 1) return LMG_ERR_ARG;
        g_fused_floor_halos = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_balance") == 0 || strcmp(key, "fused_fast") == 0) {
        if (v != 0 && v != 1) return LMG_ERR_ARG;
        (key[6] == 'b' ? g_fused_balance : g_fused_fast) = v;
        return LMG_OK;
    }
    if (strcmp(key, "fused_slow_pct") == 0) {
        if (v < 10 || v > 100) return LMG_ERR_ARG;
        g_fused_slow_pct = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_fused_tune_get(const char *key)
{
    if (strcmp(key, "fused_seg_lines") == 0) return g_fused_seg_lines;
    if (strcmp(key, "fused_pf") == 0) return g_fused_pf;
    if (strcmp(key, "fused_seg_min_lines") == 0) return g_fused_seg_min_lines;
    if (strcmp(key, "fused_seg_max_lines") == 0) return g_fused_seg_max_lines;
    if (strcmp(key, "fused_seg_lines_prol") == 0) return g_fused_seg_lines_prol;
    if (strcmp(key, "fused_seg_lines_rest") == 0) return g_fused_seg_lines_rest;
    if (strcmp(key, "fused_want_waves") == 0) return g_fused_want_waves;
    if (strcmp(key, "fused_want_waves_rest3") == 0) return g_fused_want_waves_rest3;
    if (strcmp(key, "fused_floor_halos") == 0) return g_fused_floor_halos;
    if (strcmp(key, "fused_balance") == 0) return g_fused_balance;
    if (strcmp(key, "fused_fast") == 0) return g_fused_fast;
    if (strcmp(key, "fused_slow_pct") == 0) return g_fused_slow_pct;
    return LMG_ERR_ARG;
}

extern "C" {

static int fill_args(MArgs &a, int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                     const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val, int sweeps,
                     const double *x_in, const double *b, double omega, double *x_out, double *r_out)
{
    if (n < 0 || n >= (1ll << 29) - 4096 || npat < 1 || npat > kMaxPat || (union_mask & ~0x1FFu)) return LMG_ERR_ARG;   // 32-bit byte offsets
    if (sweeps < 1 || sweeps > 3) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (n < 2) return LMG_ERR_CAPACITY;
    if (!pid || !st_val || !st_mask || !b || !x_out || x_in == x_out || r_out == x_out || (r_out && r_out == x_in))
        return LMG_ERR_ARG;
    if (line_stride < 3 || line_stride > n) return LMG_ERR_ARG;
    a.n = (int)n;
    a.W = line_stride;
    a.lines = (int)((n + line_stride - 1) / line_stride);
    a.npat = npat;
    a.umask = union_mask;
    a.strips = a.segs = a.seg_lines = 0;
    a.items = a.nb_strips = a.edge_lines = a.allow_fast = 0;
    a.segs_b = a.seg_lines_b = 1;
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
    a.hotp[0] = a.hotp[1] = -1;
    for (int k = 0; k < 9; ++k) a.hp[k] = 0.0;
    a.bc = nullptr;
    a.rpid = nullptr;
    a.rp_val = nullptr;
    a.rp_mask = nullptr;
    a.rp_npat = 0;
    a.hotr = -1;
    for (int k = 0; k < 9; ++k) a.hr[k] = 0.0;
    return 1;                                  // filled: launch
}

int lmg_stencil_smooth(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                       const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val,
                       int sweeps, const double *x_in, const double *b, double omega, double *x_out, double *r_out,
                       void *stream)
{
    MArgs a;
    const int rc = fill_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, r_out);
    if (rc != 1) return rc;
    hipStream_t st = lmg_stream(stream);
    const bool resid = r_out != nullptr, zero = x_in == nullptr;
    switch (union_mask) {
    case kMask5: return launch1<kMask5>(a, sweeps, resid, zero, st);
    case kMask9: return launch1<kMask9>(a, sweeps, resid, zero, st);
    case kMask1D: return launch1<kMask1D>(a, sweeps, resid, zero, st);
    default: return LMG_ERR_CAPACITY;        // other slot sets: run the separate sweeps
    }
}

int lmg_stencil_smooth_prolong(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                               const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val,
                               int sweeps, const double *x_in, const double *b, double omega, double *x_out,
                               int64_t n_coarse, int32_t coarse_stride, const double *e_coarse, const uint8_t *p_pid,
                               int32_t p_npat, const double *p_val, const int32_t *p_mask, const int32_t *h_hot_pairs,
                               const double *h_hot_pval, void *stream)
{
    if (!x_in || !e_coarse || !p_pid || !p_val || !p_mask || p_npat < 1 || p_npat > kMaxPat) return LMG_ERR_ARG;
    if (n_coarse < 1 || n_coarse >= (1ll << 31) || coarse_stride < 1 || coarse_stride > n_coarse) return LMG_ERR_ARG;
    if (e_coarse == x_out) return LMG_ERR_ARG;
    // the 2 x 2 window of row (y, x) starts at ((y >> 1), (x >> 1)): the coarse line stride must cover the fine one
    if ((int64_t)coarse_stride < ((int64_t)line_stride + 1) / 2) return LMG_ERR_ARG;
    MArgs a;
    const int rc = fill_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, nullptr);
    if (rc != 1) return rc;
    a.ec = e_coarse;
    a.nc = (int)n_coarse;
    a.Wc = coarse_stride;
    a.ppid = p_pid;
    a.pp_val = p_val;
    a.pp_mask = p_mask;
    a.pp_npat = p_npat;
    if (h_hot_pairs && h_hot_pval) {
        a.hotp[0] = h_hot_pairs[0];
        a.hotp[1] = h_hot_pairs[1];
        for (int k = 0; k < 9; ++k) a.hp[k] = h_hot_pval[k];
    }
    hipStream_t st = lmg_stream(stream);
    switch (union_mask) {
    case kMask5: return launch_prol<kMask5>(a, sweeps, st);
    case kMask9: return launch_prol<kMask9>(a, sweeps, st);
    default: return LMG_ERR_CAPACITY;
    }
}

int lmg_stencil_smooth_restrict(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                                const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val,
                                int sweeps, const double *x_in, const double *b, double omega, double *x_out,
                                int64_t n_coarse, int32_t coarse_stride, double *b_coarse, const uint8_t *r_pid,
                                int32_t r_npat, const double *r_val, const int32_t *r_mask, int32_t hot_r,
                                const double *h_hot_rval, void *stream)
{
    if (!b_coarse || !r_pid || !r_val || !r_mask || r_npat < 1 || r_npat > kMaxPat) return LMG_ERR_ARG;
    if (n_coarse < 1 || n_coarse >= (1ll << 28) || coarse_stride < 1 || coarse_stride > n_coarse) return LMG_ERR_ARG;
    if ((const double *)b_coarse == x_in || b_coarse == x_out || (const double *)b_coarse == b) return LMG_ERR_ARG;
    // row (Y, X) of R sits on the fine node (2 Y, 2 X): every such node of the fine grid must have its coarse row
    const int64_t lines = n > 0 ? (n + line_stride - 1) / line_stride : 0;
    // -- and nothing else: the pass only writes b_coarse under those nodes, a larger coarse grid would keep stale rows
    if ((int64_t)coarse_stride != ((int64_t)line_stride + 1) / 2 || (n % line_stride) != 0 || n_coarse != ((lines + 1) / 2) * coarse_stride)
        return LMG_ERR_ARG;
    MArgs a;
    const int rc = fill_args(a, n, line_stride, pid, npat, st_val, st_mask, union_mask, hot_pattern, h_hot_val, sweeps, x_in, b,
                             omega, x_out, nullptr);
    if (rc != 1) return rc;
    a.bc = b_coarse;
    a.nc = (int)n_coarse;
    a.Wc = coarse_stride;
    a.rpid = r_pid;
    a.rp_val = r_val;
    a.rp_mask = r_mask;
    a.rp_npat = r_npat;
    if (hot_r >= 0 && hot_r < r_npat && h_hot_rval) {
        a.hotr = hot_r;
        for (int k = 0; k < 9; ++k) a.hr[k] = h_hot_rval[k];
    }
    hipStream_t st = lmg_stream(stream);
    const bool zero = x_in == nullptr;
    switch (union_mask) {
    case kMask5: return launch_rest<kMask5>(a, sweeps, zero, st);
    case kMask9: return launch_rest<kMask9>(a, sweeps, zero, st);
    default: return LMG_ERR_CAPACITY;
    }
}

int lmg_stencil_smooth_supported(uint32_t union_mask)
{
    return union_mask == kMask5 || union_mask == kMask9 || union_mask == kMask1D;
}

int lmg_stencil_smooth_prolong_supported(uint32_t union_mask)
{
    return union_mask == kMask5 || union_mask == kMask9;
}

}  // extern "C"
