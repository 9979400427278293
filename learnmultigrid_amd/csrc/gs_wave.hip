// Exact forward (lexicographic) Gauss-Seidel on grid-stencil matrices for gfx950: a pipelined wavefront
// in registers instead of one dependent kernel launch (or workgroup barrier) per anti-diagonal.
//
// pyamg's gauss_seidel (Multigrid.py:88, :121) relaxes row i with the NEW values of all columns j < i and the
// OLD values of all j > i.  On a grid-stencil matrix (format of stencil.hip: every entry at column - row =
// c*W + d, c, d in {-1, 0, 1}) with no coupling across line ends, row (y, x) needs the new values of
// (y-1, x-1 .. x+1) and (y, x-1): a chain of W + (#lines) * SK dependent updates, SK = 2 when the upper-right
// neighbour (y-1, x+1) is coupled, else 1.  That chain is the critical path of ANY exact implementation;
// what can be removed is everything else on it.  csr_gs_schedule pays a launch (5.3 us) or a workgroup
// barrier plus an L2 round trip (2.4 us) per anti-diagonal.  Here:
//   * a WAVE owns a band of 64 consecutive lines, lane l the line y0 + l, and at step t lane l relaxes
//     column t - SK*l: the new value a lane needs from the line above was produced by lane l-1 one step
//     earlier and arrives through a DPP wave shift; its own previous result is in a register; the old values
//     of its own line and of the line below, b and the pattern id are loaded PF steps ahead.  One step is
//     the arithmetic of one row (a handful of multiply-adds and one IEEE division), ~0.15 us.
//   * bands are pipelined against each other through memory: the last lane of a band stores its results
//     write-through (sc1) and publishes "columns done" a few steps later behind a COUNTED s_waitcnt (the
//     stores of that many steps ago have completed; nothing stalls), lane 0 of the next band prefetches that
//     counter and the values above its line with sc1 loads and only spins when it has caught up.  A band
//     therefore trails its predecessor by 64*SK steps plus two memory latencies.
//   * bands are handed out by an atomic ticket, so a band only ever waits for one that is already running
//     (no assumption about the dispatch order of workgroups).
// Same row arithmetic in the same order as gs_update_row (rsum over the off-diagonal entries in column order,
// (b - rsum) / diag, rows with a zero diagonal untouched): bit-identical to the level-scheduled sweep and to
// the CPU oracle.  4097^2: one sweep ~1.5 ms instead of 43.7 ms (8191 launches); 513^2 ~0.25 ms instead of 2.5.
#include <math.h>
#include <string.h>
#include "lmg_common.hpp"

namespace {

constexpr int kMaxPat = 64;
constexpr int kPF = 5;                      // iterations (of two steps) between issuing a load and using its value
#ifndef LMG_GS_HYST
#define LMG_GS_HYST 2          // (8: 8.12 ms per sweep at 4097^2, 4: 7.66, 2: 7.57; 513^2: 0.93 / 0.85 / 0.82)
#endif
constexpr int kPubDelay = 6;                // iterations between a result store and the progress that covers it: the
                                            // counted wait in front of the progress store also covers every LOAD issued
                                            // before that store, so it must not be shorter than the prefetch distance
constexpr int kVmOpsBase = 11;              // vector-memory instructions per iteration: 6 loads, 2 + 2 result stores, progress
                                            // (+ 1 load with several sweeps per launch)
constexpr int kMaxSweeps = 4;               // sweeps pipelined behind each other in one launch
constexpr unsigned kMask5 = 0x0BAu, kMask9 = 0x1FFu, kMask7 = 0x1BBu, kMask1D = 0x038u;
constexpr unsigned kOOB = 0xFFFFFFF0u;      // buffer offset beyond any num_records: the access is dropped / reads 0
constexpr int kSc1 = 16;                    // buffer cache policy: sc1 (write-through / L1 bypass, agent scope)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

struct GArgs {
    int n, W, lines, nbands, npat;
    const unsigned char *pid;
    const double *st_val;
    const int *st_mask;
    double *x;
    const double *b;
    int hot;                                // interior pattern (all union slots, non-zero diagonal) or -1
    double hot_val[9];
    double hot_rcp;                         // 1 / hot_val[4] when that diagonal is a power of two (the quotient is then a product, bit for bit), else 0
    int sweeps;                             // sweeps of this launch (<= kMaxSweeps), pipelined: see gs_wavefront_kernel
    int *work;                              // [0] error flag, [1] ticket, [2] where the other lanes "publish",
                                            // [3 + s * nbands + k] columns done on the last line of band k in sweep s
};

__device__ __forceinline__ double dpp_lower(double src)      // lane i <- lane i-1, lane 0 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lo2(const u4 &v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double hi2(const u4 &v) { return __hiloint2double((int)v.w, (int)v.z); }

// elements i, i+1 of a double array through a buffer resource: out-of-range pairs read 0; the two pairs that
// straddle the first / last element are read one element further in and sorted out by the consumer
// (`straddle`: +1 = i == n-1, -1 = i == -1), so that no access is ever PARTLY out of range
__device__ __forceinline__ unsigned pair_off(int64_t i, int n)
{
    const int64_t j = i + (i == -1 ? 1 : 0) - (i == (int64_t)n - 1 ? 1 : 0);
    return (j >= 0 && j + 1 < n) ? (unsigned)j * 8u : kOOB;
}

struct Ahead {          // what iteration it + kPF needs (two columns), on its way through memory
    u4 own, down, up, b;
    int pid2, flag, flag_old;
};

// MULTI: several sweeps in ONE launch, pipelined behind each other.  Tickets run over (sweep, band) in row-major order;
// band b of sweep s additionally trails band b + 1 of sweep s - 1 (the last band: band b itself): once that one has
// finished columns <= c on ALL its lines, the old values this band reads there -- its own lines and the line below --
// are the previous sweep's final ones, and nobody will read what this band overwrites.  Both dependencies hold
// smaller tickets, so a band only ever waits for bands that are running or done.  The pipeline fill (64 * SK steps per
// band, two thirds of a sweep at 4097^2) is paid once per launch instead of once per sweep.  All result stores are
// write-through and all loads of x bypass the caches in this mode (other workgroups read / wrote them in this launch).
template <unsigned UM, bool MULTI>
__global__ void __launch_bounds__(64) gs_wavefront_kernel(GArgs a)
{
    constexpr int SK = (UM & 4u) ? 2 : 1;
    constexpr int kXPol = MULTI ? kSc1 : 0;                       // cache policy of the loads of x
    __shared__ double s_val[kMaxPat * 9];
    __shared__ int s_mask[kMaxPat];
    __shared__ int s_band;
    const int lane = threadIdx.x;
    for (int i = lane; i < a.npat * 9; i += 64) s_val[i] = a.st_val[i];
    for (int i = lane; i < a.npat; i += 64) s_mask[i] = a.st_mask[i];
    if (lane == 0) s_band = atomicAdd(&a.work[1], 1);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_band);
    if (ticket >= a.nbands * a.sweeps) return;
    const int sweep = ticket / a.nbands, band = ticket - sweep * a.nbands;

    const int n = a.n, W = a.W;
    const int y = band * 64 + lane;
    const bool line_ok = y < a.lines;
    const int64_t base = (int64_t)y * W;
    const int last_lane = min(63, a.lines - 1 - band * 64);          // lane of the band's last line
    int *const prog = a.work + 3 + sweep * a.nbands;
    int *prog_mine = lane == last_lane ? prog + band : a.work + 2;    // other lanes: a dummy word
    // (counters that are read but ignored point at the error flag, a word nobody writes: the dummy word [2] is
    // stored to by every wave in every iteration, and a load behind that traffic holds up the in-order vmcnt)
    const int *prog_prev = band > 0 ? prog + band - 1 : a.work;
    const bool has_old = MULTI && sweep > 0;
    const int *prog_old = has_old ? prog - a.nbands + min(band + 1, a.nbands - 1) : a.work;
    const int ITER = (W + SK * 63 + 1) / 2 + 1;                       // iterations of a band (two steps each)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, (int)((unsigned)n * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)a.b, 0, (int)((unsigned)n * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)a.pid, 0, n, 0x00020000);

    // loads of iteration itf (columns xf, xf + 1 with xf = 2 itf - SK lane): always the same six instructions
    auto fetch = [&](int itf, Ahead &A) {
        const int xf = 2 * itf - SK * lane;
        const int64_t i = base + xf;
        A.own = __builtin_amdgcn_raw_buffer_load_b128(rs_x, pair_off(i + 1, n), 0, kXPol);      // old (y, xf+1), (y, xf+2)
        A.down = __builtin_amdgcn_raw_buffer_load_b128(rs_x, pair_off(i + W + 1, n), 0, kXPol); // old (y+1, xf+1), (y+1, xf+2)
        A.b = __builtin_amdgcn_raw_buffer_load_b128(rs_b, pair_off(i, n), 0, 0);
        {
            const int64_t j = i + (i == -1 ? 1 : 0) - (i == (int64_t)n - 1 ? 1 : 0);
            A.pid2 = (int)__builtin_amdgcn_raw_buffer_load_b16(rs_p, (j >= 0 && j + 1 < n) ? (unsigned)j : kOOB, 0, 0);
        }
        // lane 0: new (y-1, xf+SK-1), (y-1, xf+SK) of the previous band, write-through data -> L1 bypass
        A.up = __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane == 0 ? pair_off(i - W + (SK - 1), n) : kOOB, 0, kSc1);
        A.flag = __hip_atomic_load(prog_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        A.flag_old = MULTI ? __hip_atomic_load(prog_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    };
    // lane 0's read of the line above is only legal once the previous band has published those columns
    int spin_budget = 1 << 22;
    auto wait_for = [&](const int *counter, int need, int flag_seen) {
        int f = __builtin_amdgcn_readfirstlane(flag_seen);
        if (f >= need) return;
        // Caught up with the previous band: wait until it is comfortably ahead (or done), not just one column --
        // a band that trails by exactly the dependency distance would come back here every iteration and advance
        // at one memory round trip per iteration (measured: 1.5 us instead of 0.3)
        need = min(W, need + LMG_GS_HYST * kPF);
        // (bounded: the previous band holds an earlier ticket, so it is running or done and its counter only
        // grows; the budget -- a few seconds per band in total -- turns a protocol bug into a wrong result with
        // an error flag instead of a hung GPU)
        while (f < need && spin_budget > 0) {
            --spin_budget;
            __builtin_amdgcn_s_sleep(2);
            f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (f < need && lane == 0) a.work[0] = 1;
    };
    auto wait_for_prev = [&](int itf, int flag_seen, int flag_old_seen) {
        // the line above: columns 0 .. 2 itf + SK of the previous band done
        if (band > 0) wait_for(prog_prev, min(W, 2 * itf + SK + 1), flag_seen);
        // the previous sweep: columns 0 .. 2 itf + 2 of this band's lines and of the line below final
        if (has_old) wait_for(prog_old, min(W, 2 * itf + 3), flag_old_seen);
    };

    double U0 = 0.0, U1 = 0.0, U2 = 0.0;      // new values of the line above at columns x-1, x, x+1
    double D0 = 0.0, D1 = 0.0, D2 = 0.0;      // old values of the line below
    double O0 = 0.0, O1 = 0.0;                // old values of the own line at columns x, x+1
    double R = 0.0;                           // own result of the previous step = new (y, x-1)
    Ahead ring[kPF];
    int flag_seen = 0, flag_old_seen = 0;
    // prologue: the first kPF iterations' loads (lane 0's line-above loads need the previous band first)
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
        wait_for_prev(u, flag_seen, flag_old_seen);
        fetch(u, ring[u]);
    }
    {   // windows just before the first step (column x0 = -SK lane): what the first shift moves into place
        const int64_t i0 = base - SK * lane;
        auto one = [&](int64_t i) {
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (i >= 0 && i < n) ? (unsigned)i * 8u : kOOB, 0, kSc1);
            return __hiloint2double((int)v.y, (int)v.x);
        };
        O1 = one(i0);                         // old (y, x0)
        D2 = one(i0 + W);                     // old (y+1, x0)
        D1 = one(i0 + W - 1);                 // old (y+1, x0-1)
        if (SK == 2 && lane == 0 && band > 0) U2 = one(i0 - W);             // new (y-1, 0): becomes U1 at column 0
    }

    // hot pattern (the interior row): values in scalar registers, used when every lane of the wave relaxes
    // such a row in this step -- no LDS reads, no selects, the same products and sums in the same order
    const int hot = a.hot;
    double hv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) hv[q] = a.hot_val[q];

    // one step = one column: windows move one to the right, then row (y, x) is relaxed
    auto step = [&](int x, double up_top, double down_new, double own_new, double bval, int p_in) -> double {
        U0 = U1;
        U1 = U2;
        if (SK == 2) U2 = up_top;
        else U1 = up_top;
        D0 = D1;
        D1 = D2;
        D2 = down_new;
        O0 = O1;
        O1 = own_new;
        const int64_t i = base + x;
        const bool act = line_ok && x >= 0 && x < W && i < n;
        double xn;
        if (__all(act && p_in == hot)) {                       // wave-uniform
            double rsum = 0.0;
            if ((UM >> 0) & 1u) rsum = rsum + hv[0] * U0;
            if ((UM >> 1) & 1u) rsum = rsum + hv[1] * U1;
            if ((UM >> 2) & 1u) rsum = rsum + hv[2] * U2;
            if ((UM >> 3) & 1u) rsum = rsum + hv[3] * R;
            if ((UM >> 5) & 1u) rsum = rsum + hv[5] * O1;
            if ((UM >> 6) & 1u) rsum = rsum + hv[6] * D0;
            if ((UM >> 7) & 1u) rsum = rsum + hv[7] * D1;
            if ((UM >> 8) & 1u) rsum = rsum + hv[8] * D2;
            xn = (bval - rsum) / hv[4];
            R = xn;
        } else {
            const int p = act ? p_in : 0;
            const int m = act ? s_mask[p] : 0;
            double vv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) vv[q] = ((UM >> q) & 1u) ? s_val[p * 9 + q] : 0.0;     // all reads up front
            double rsum = 0.0;
            double t_;
            if ((UM >> 0) & 1u) { t_ = rsum + vv[0] * U0; rsum = ((m >> 0) & 1) ? t_ : rsum; }
            if ((UM >> 1) & 1u) { t_ = rsum + vv[1] * U1; rsum = ((m >> 1) & 1) ? t_ : rsum; }
            if ((UM >> 2) & 1u) { t_ = rsum + vv[2] * U2; rsum = ((m >> 2) & 1) ? t_ : rsum; }
            if ((UM >> 3) & 1u) { t_ = rsum + vv[3] * R; rsum = ((m >> 3) & 1) ? t_ : rsum; }
            if ((UM >> 5) & 1u) { t_ = rsum + vv[5] * O1; rsum = ((m >> 5) & 1) ? t_ : rsum; }
            if ((UM >> 6) & 1u) { t_ = rsum + vv[6] * D0; rsum = ((m >> 6) & 1) ? t_ : rsum; }
            if ((UM >> 7) & 1u) { t_ = rsum + vv[7] * D1; rsum = ((m >> 7) & 1) ? t_ : rsum; }
            if ((UM >> 8) & 1u) { t_ = rsum + vv[8] * D2; rsum = ((m >> 8) & 1) ? t_ : rsum; }
            const double diag = ((m >> 4) & 1) ? vv[4] : 0.0;
            const double q_ = (bval - rsum) / (diag != 0.0 ? diag : 1.0);
            xn = diag != 0.0 ? q_ : O0;
            R = act ? xn : R;
        }
        return xn;
    };

    for (int ib = 0; ib < ITER; ib += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const int it = ib + u;
            const int x = 2 * it - SK * lane;             // columns x, x + 1 in this iteration
            const Ahead cur = ring[u];
            flag_seen = cur.flag;
            flag_old_seen = cur.flag_old;
            const int64_t i = base + x;
            // which half of a pair is which (only the pairs straddling element 0 / n-1 are special)
            auto first = [&](const u4 &v, int64_t ii) { return ii == (int64_t)n - 1 ? hi2(v) : lo2(v); };
            auto second = [&](const u4 &v, int64_t ii) { return ii == -1 ? lo2(v) : hi2(v); };
            const int pa = (i == (int64_t)n - 1) ? (cur.pid2 >> 8) & 0xff : cur.pid2 & 0xff;
            const int pb = (i == -1) ? cur.pid2 & 0xff : (cur.pid2 >> 8) & 0xff;
            // ---- next loads ----------------------------------------------------------------------------
            wait_for_prev(it + kPF, flag_seen, flag_old_seen);
            fetch(it + kPF, ring[u]);
            // ---- two steps -----------------------------------------------------------------------------
            // (the wave shifts run with all lanes enabled: a DPP read from a lane that a branch has switched off
            // returns the destination's old value, not the neighbour's)
            const double inA = dpp_lower(R);
            const double upA = lane == 0 ? first(cur.up, i - W + (SK - 1)) : inA;
            const double xa = step(x, upA, first(cur.down, i + W + 1), first(cur.own, i + 1), first(cur.b, i), pa);
            const double inB = dpp_lower(R);
            const double upB = lane == 0 ? second(cur.up, i - W + (SK - 1)) : inB;
            const double xb = step(x + 1, upB, second(cur.down, i + W + 1), second(cur.own, i + 1), second(cur.b, i), pb);
            // ---- results: one 16-byte and one 8-byte store, disabled ones out of range.  Only the band's LAST line is
            // read by another wave during this launch (the next band's lane 0), so only that lane stores write-through
            // (sc1) -- 64 write-through lines per instruction cost 2 us per iteration (SQ_WAIT_ANY 70 %: every later
            // load is counted behind them in vmcnt); all other lanes use plain stores, flushed at the kernel boundary.
            const bool actA = line_ok && x >= 0 && x < W && i < n, actB = line_ok && x + 1 >= 0 && x + 1 < W && i + 1 < n;
            const bool shared = MULTI || lane == last_lane;
            u4 v4;
            v4.x = (unsigned)__double2loint(xa);
            v4.y = (unsigned)__double2hiint(xa);
            v4.z = (unsigned)__double2loint(xb);
            v4.w = (unsigned)__double2hiint(xb);
            const unsigned off16 = (actA && actB) ? (unsigned)i * 8u : kOOB;
            u2 v2;
            v2.x = actA ? v4.x : v4.z;
            v2.y = actA ? v4.y : v4.w;
            const unsigned off8 = (actA != actB) ? (unsigned)(actA ? i : i + 1) * 8u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(v4, rs_x, shared ? kOOB : off16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(v2, rs_x, shared ? kOOB : off8, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v4, rs_x, shared ? off16 : kOOB, 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b64(v2, rs_x, shared ? off8 : kOOB, 0, kSc1);
            // ---- publish: the result stores of kPubDelay iterations ago have completed ----------------------
            {
                constexpr int kVmOpsPerIter = kVmOpsBase + (MULTI ? 1 : 0);
                constexpr int N = kVmOpsPerIter * (kPubDelay - 1);
                static_assert(N < 64 && kVmOpsPerIter * kPF < 64, "vmcnt range");
                __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
                // last line: after iteration it' its columns 0 .. 2 it' + 1 - SK last_lane are done
                const int done = min(W, 2 * (it - kPubDelay) + 2 - SK * last_lane);
                __hip_atomic_store(prog_mine, max(done, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // everything of this band is done once all stores have completed
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    __hip_atomic_store(prog_mine, W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the same sweep with the band's data staged through LDS --------------------------------------------------------
// gs_wavefront_kernel issues, per iteration, five vector-memory instructions whose 64 lanes sit in 64 different grid
// lines (one cache line each): ~1 us per anti-diagonal, all of it address processing.  Here the band's lines travel in
// CHUNKS of 16 columns: every chunk is brought in by LDS-DMA (global_load_lds: 8 grid lines x 128 B per instruction,
// each lane group a whole cache line), a
// ring of 8 chunk slots (65 lines of x, 64 of b, the pattern ids, the last line of the band above) holds the window the
// 64 skewed lanes work in, results are written back INTO the tile and a finished chunk leaves by 8 coalesced stores.
// A step then touches LDS only; the band above is waited for once per chunk (its progress counter now counts flushed
// chunks).  5- and 7-point operators: lane l relaxes column t - l, 64-line bands; 9-point operators (upper-right slot): column
// t - 2 l, 32-line bands (see SK / NL below).  Same row arithmetic in the same order as gs_wavefront_kernel: bit-identical.
constexpr int kCW = 16;                                    // columns per chunk
constexpr int kNCH = 8;                                    // ring slots
constexpr int kXSBytes = 72 * 128;                         // x: 9 groups of 8 lines (65 used) x 16 columns
constexpr int kBSBytes = 64 * 128;
constexpr int kPSBytes = 5 * 256;                          // pattern ids: 5 aligned dwords per line, [dword][line]
constexpr int kUSBytes = 128;                              // the last line of the band above
constexpr int kSlotBytes = kXSBytes + kBSBytes + kPSBytes + kUSBytes;
constexpr int kLead = 2;                                   // chunks requested ahead of lane 0
constexpr int kFlushLag = 5;                               // chunk p - 5 is complete (lane 63 has left it) when lane 0 enters chunk p

#define LMG_GLDS(gptr, lptr, bytes, aux)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                  \
                                     (__attribute__((address_space(3))) void *)(lptr), bytes, 0, aux)

// MULTI: several sweeps in one launch, pipelined like gs_wavefront_kernel<UM, true> (tickets over (sweep, band) in row-major order;
// band b of sweep s asks for chunk c once band b + 1 of sweep s - 1 -- the last band: band b itself -- has FLUSHED chunk c: then its
// own 64 lines and the line below hold the previous sweep's final values there, and nobody will read what this band overwrites).
// Every line is then stored write-through and x comes in past the caches, like the line above always does.
template <unsigned UM, bool MULTI = false>
__global__ void __launch_bounds__(64) gs_band_lds_kernel(GArgs a)
{
    // SK = 2 (the upper-right slot is coupled: 9-point operators): lane l relaxes column t - 2 l, and a band has 32 lines (lanes
    // 32 .. 63 idle) so that the window of the skewed lanes still spans 62 columns = the 8-slot ring; the slots keep their
    // 64-line layout.  A lone wave pays per instruction, not per lane: the idle half costs nothing.
    constexpr int SK = (UM & 4u) ? 2 : 1;
    constexpr int NL = SK == 2 ? 32 : 64;                             // lines of a band
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[kNCH * kSlotBytes + LMG_WAVE * 8];   // + one dump slot per lane
    __shared__ double s_val[kMaxPat * 9];
    __shared__ int s_mask[kMaxPat];
    __shared__ int s_band;
    const int lane = threadIdx.x;
    for (int i = lane; i < a.npat * 9; i += 64) s_val[i] = a.st_val[i];
    for (int i = lane; i < a.npat; i += 64) s_mask[i] = a.st_mask[i];
    if (lane == 0) s_band = atomicAdd(&a.work[1], 1);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_band);
    if (ticket >= a.nbands * (MULTI ? a.sweeps : 1)) return;
    const int sweep = MULTI ? ticket / a.nbands : 0, band = ticket - sweep * a.nbands;

    const int n = a.n, W = a.W;
    const int y0 = band * NL;
    const int y = y0 + lane;
    const bool line_ok = lane < NL && y < a.lines;
    const int last_lane = min(NL - 1, a.lines - 1 - y0);
    const int cmax = (W - 1) / kCW;                                   // last chunk
    int *const prog = a.work + 3 + sweep * a.nbands;
    const int *prog_prev = band > 0 ? prog + band - 1 : a.work;
    const bool has_old = MULTI && sweep > 0;
    const int *prog_old = has_old ? prog - a.nbands + min(band + 1, a.nbands - 1) : a.work;
    constexpr int kXPol = MULTI ? kSc1 : 0;                           // cache policy of the band's own loads of x
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, (int)((unsigned)n * 8u), 0x00020000);

    // ---- LDS addressing: plain rows of 128 B.  Lane l reads row l at column t - l: the SKEW of the wavefront spreads the
    // 64 lanes over the banks by itself (30 l + 2 t mod 64 dwords: conflict-free; an XOR swizzle of the pieces made it 5x worse)
    const unsigned rowoff = (unsigned)lane * 128u, frow = 0u;
    const unsigned rowoff1 = (unsigned)(lane + 1) * 128u, frow1 = 0u;
    auto taddr = [&](unsigned roff, unsigned f, int col) -> unsigned {
        return (((unsigned)col >> 4) & (kNCH - 1)) * kSlotBytes + roff + ((((unsigned)col >> 1) & 7u) ^ f) * 16u + ((unsigned)col & 1u) * 8u;
    };
    const unsigned palign = (unsigned)(((int64_t)y * W) & 3);         // byte misalignment of the lane's line in the id array
    auto ldsd = [&](unsigned off) -> double { return *reinterpret_cast<const double *>(s_tile + off); };

    // ---- chunk c comes in: 9 + 8 instructions of 8 lines x 128 B, 5 dwords of ids per line, the line above -------------
    auto issue_up = [&](int c) {                                      // the last line of the band above, columns of chunk c
        if (c > cmax || band == 0) return;
        if (lane < 8) {
            const int64_t idx = (int64_t)(y0 - 1) * W + c * kCW + 2 * lane;
            LMG_GLDS(a.x + idx, s_tile + (c & (kNCH - 1)) * kSlotBytes + kXSBytes + kBSBytes + kPSBytes, 16, kSc1);   // written write-through there
        }
    };
    auto issue_chunk = [&](int c, bool with_up) {
        if (c > cmax) return;                                         // uniform
        unsigned char *sb = s_tile + (c & (kNCH - 1)) * kSlotBytes;
        const int col0 = c * kCW;
        const int r = lane >> 3, q = lane & 7;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int row = 8 * g + r;
            const int pce = q;
            const int64_t idx = (int64_t)(y0 + row) * W + col0 + 2 * pce;
            // (a piece that straddles the end of the vector is 16-byte aligned: the vectors are, and n - 1 is even there)
            if (row <= NL && y0 + row < a.lines && idx < n) LMG_GLDS(a.x + idx, sb + g * 1024, 16, kXPol);
            if (row < NL && y0 + row < a.lines && idx < n) LMG_GLDS(a.b + idx, sb + kXSBytes + g * 1024, 16, 0);
        }
        if (line_ok) {
            const unsigned char *src = a.pid + ((int64_t)y * W - palign) + col0;
#pragma unroll
            for (int k = 0; k < 5; ++k) LMG_GLDS(src + 4 * k, sb + kXSBytes + kBSBytes + k * 256, 4, 0);
        }
        if (with_up) issue_up(c);
    };
    // ---- chunk fc leaves: 8 lines x 128 B per store instruction; the band's last line write-through ----------------------
    auto flush_chunk = [&](int fc) {
        if (fc < 0 || fc > cmax) return;
        const unsigned char *sb = s_tile + (fc & (kNCH - 1)) * kSlotBytes;
        const int col0 = fc * kCW;
        const int r = lane >> 3, q = lane & 7;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (8 * g > last_lane) break;                             // uniform
            const int row = 8 * g + r;
            const int col = col0 + 2 * q;
            const u4 v = *reinterpret_cast<const u4 *>(sb + g * 1024 + lane * 16);
            const int64_t i = (int64_t)(y0 + row) * W + col;
            const bool rok = row <= last_lane;
            const unsigned off16 = (rok && col + 1 < W) ? (unsigned)i * 8u : kOOB;
            const unsigned off8 = (rok && col + 1 == W) ? (unsigned)i * 8u : kOOB;
            u2 v2;
            v2.x = v.x;
            v2.y = v.y;
            if (MULTI || g == (last_lane >> 3)) {                     // uniform: the group with the line the next band reads
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, off16, 0, kSc1);
                __builtin_amdgcn_raw_buffer_store_b64(v2, rs_x, off8, 0, kSc1);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, off16, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(v2, rs_x, off8, 0, 0);
            }
        }
    };
    int spin_budget = 1 << 22;
    int flag_seen = 0;                                                // the band above's progress as last seen (polled ahead of its use)
    auto wait_prev = [&](int c) {                                     // columns of chunk c of the line above final?
        if (band == 0 || c > cmax) return;
        const int need = min(W, (c + 1) * kCW);
        int f = __builtin_amdgcn_readfirstlane(flag_seen);
        if (f >= need) return;
        f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        while (f < need && spin_budget > 0) {
            --spin_budget;
            __builtin_amdgcn_s_sleep(4);
            f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (f < need && lane == 0) a.work[0] = 1;
    };
    int flag_old_seen = 0;
    auto wait_old = [&](int c) {                                      // chunk c of the previous sweep final (own lines, line below)?
        if (!has_old || c > cmax) return;
        const int need = min(W, (c + 1) * kCW);
        int f = __builtin_amdgcn_readfirstlane(flag_old_seen);
        if (f >= need) return;
        f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        while (f < need && spin_budget > 0) {
            --spin_budget;
            __builtin_amdgcn_s_sleep(4);
            f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (f < need && lane == 0) a.work[0] = 1;
    };
    auto publish = [&](int done_cols) {
        if (lane == 0) __hip_atomic_store(prog + band, done_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- what a step needs, read from the tile one step ahead ---------------------------------------------------------------
    // (the b tile has the layout of the x tile, kXSBytes further on: the address of b (y, c) is the address x (y, c) had a step ago)
    struct In { double own1, down2, b, up; int p; };
    unsigned own_addr_prev = 0;                                       // LDS address of x (y, c) for the column c read as "own1" a step ago
    // No conditionals: every address lies inside the ring whatever x is, a value read for a column outside the line (or for a
    // lane that has not started / has finished) is never used -- the row patterns there have no such slot (no coupling across
    // line ends: StencilTwin.gs_ok), inactive lanes compute nothing that is kept.
    auto read_inputs = [&](int x) -> In {                             // for the step that relaxes column x of the lane's line
        In I;
        const unsigned c1 = (unsigned)(x + 1);
        const unsigned s1 = ((c1 >> 4) & (kNCH - 1)) * kSlotBytes, pc1 = (c1 >> 1) & 7u, h1 = (c1 & 1u) * 8u;
        const unsigned own_addr = s1 + rowoff + pc1 * 16u + h1;
        I.own1 = ldsd(own_addr);
        I.down2 = ldsd(own_addr + 128u);
        I.b = ldsd(own_addr_prev + kXSBytes);
        own_addr_prev = own_addr;
        const unsigned sbo = (((unsigned)x >> 4) & (kNCH - 1)) * kSlotBytes;
        const unsigned o = palign + ((unsigned)x & 15u);
        I.p = (int)s_tile[sbo + kXSBytes + kBSBytes + (o >> 2) * 256u + (unsigned)lane * 4u + (o & 3u)];
        if (SK == 1) {
            I.up = ldsd(sbo + kXSBytes + kBSBytes + kPSBytes + ((unsigned)x & 15u) * 8u);
        } else {                                                      // the line above at column x + 1 (its upper-right neighbour)
            const unsigned xu = (unsigned)(x + 1);
            I.up = ldsd(((xu >> 4) & (kNCH - 1)) * kSlotBytes + kXSBytes + kBSBytes + kPSBytes + (xu & 15u) * 8u);
        }
        return I;
    };

    const int hot = a.hot;
    double hv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) hv[q] = a.hot_val[q];
    // a diagonal that is a power of two: t / d == t * (1 / d) bit for bit (both round the same exact number), so the
    // interior rows skip the division -- ~25 dependent instructions on the critical path of every step
    const double hrc = a.hot_rcp;
    const bool hpow2 = hrc != 0.0;

    // prologue: the first kLead chunks (the line above: one chunk less ahead, see below)
    for (int c = 0; c < kLead; ++c) {
        wait_old(c);
        issue_chunk(c, false);                                      // (the line above depends on another band of this sweep)
    }
    for (int c = 0; c < kLead - 1; ++c) {
        wait_prev(c);
        issue_up(c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    double U0 = 0.0, U1 = 0.0, U2 = 0.0;      // new values of the line above at columns x-1, x (SK = 2: and x+1)
    double D0 = 0.0, D1 = 0.0, D2 = 0.0;      // old values of the line below at x-1, x, x+1
    double O0 = 0.0, O1 = 0.0;                // old values of the own line at x, x+1
    double R = 0.0;                           // own result of the previous step = new (y, x-1)
    int flushed = -1;                         // last chunk whose stores have completed and been published
    const unsigned w_dump = (unsigned)(kNCH * kSlotBytes) + (unsigned)lane * 8u;
    unsigned w_addr = w_dump;                 // the previous step's result, written to the tile at the top of the next step
    unsigned a_x = w_dump;                    // tile address of the element this step relaxes
    double w_val = 0.0;
    int issued_flush = -1;
    const int T_end = W + SK * last_lane;     // steps: the last line relaxes column W - 1 at step W - 1 + SK last_lane
    // One chunk period (16 steps) per trip, unrolled: the position in the period is a compile-time number -- no per-step tests for
    // the periodic work, the five windows rotate by renaming, LDS reads of later steps can move up.  The march starts a period
    // early (steps -16 .. -2 are idle for every lane: x < -1) so that the periods stay aligned with the chunks.
    In nxt = read_inputs(-kCW - SK * lane);
    // every lane needs old (y, 0) / (y+1, 0) when it starts: the step before its first one reads them (x = -1)
    for (int tb = -kCW; tb < T_end; tb += kCW) {
#pragma unroll
      for (int u = 0; u < kCW; ++u) {
        const int t = tb + u;
        if (u == 0 && t >= 0) {
            // lane 0 enters chunk p: chunk p - 5 is complete and leaves; chunk p + 2 is requested -- and the line above of
            // chunk p + 1 (one chunk less ahead: what this band waits for is the band above, so it asks as late as it can)
            const int p = t >> 4;
            if (p - kFlushLag >= 0 && p - kFlushLag <= cmax) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                flush_chunk(p - kFlushLag);
                issued_flush = p - kFlushLag;
            }
            wait_prev(p + kLead - 1);
            wait_old(p + kLead);
            issue_chunk(p + kLead, false);
            issue_up(p + kLead - 1);
        } else if (u == 6 && t >= 0) {
            // six steps later everything requested at the top of the period has landed and the flush stores are done:
            // publish them (the band below waits for exactly this), and look at the band above now for the next period
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (issued_flush > flushed) {
                flushed = issued_flush;
                publish(min(W, (flushed + 1) * kCW));
            }
            flag_seen = __hip_atomic_load(prog_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MULTI) flag_old_seen = __hip_atomic_load(prog_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const In cur = nxt;
        const int x = t - SK * lane;
        *reinterpret_cast<double *>(s_tile + w_addr) = w_val;               // (the previous step's result; a lane without one writes its dump slot)
        const unsigned a_x1 = own_addr_prev;                           // tile address of (y, x + 1): what the call below reads b through
        nxt = read_inputs(x + 1);
        // ---- windows move one column to the right -----------------------------------------------------------------------
        const double inU = dpp_lower(R);                               // lane l-1's result of the previous step = new (y-1, x)
        U0 = U1;
        if (SK == 2) {
            U1 = U2;
            U2 = (lane == 0) ? cur.up : inU;                           // lane l-1 relaxed column x + 1 a step ago
        } else {
            U1 = (lane == 0) ? cur.up : inU;
        }
        D0 = D1;
        D1 = D2;
        D2 = cur.down2;
        O0 = O1;
        O1 = cur.own1;
        const bool act = line_ok && x >= 0 && x < W;
        double xn;
        if (__all(act && cur.p == hot)) {                              // wave-uniform
            double rsum = 0.0;
            if ((UM >> 0) & 1u) rsum = rsum + hv[0] * U0;
            if ((UM >> 1) & 1u) rsum = rsum + hv[1] * U1;
            if ((UM >> 2) & 1u) rsum = rsum + hv[2] * U2;
            if ((UM >> 3) & 1u) rsum = rsum + hv[3] * R;
            if ((UM >> 5) & 1u) rsum = rsum + hv[5] * O1;
            if ((UM >> 6) & 1u) rsum = rsum + hv[6] * D0;
            if ((UM >> 7) & 1u) rsum = rsum + hv[7] * D1;
            if ((UM >> 8) & 1u) rsum = rsum + hv[8] * D2;
            xn = hpow2 ? (cur.b - rsum) * hrc : (cur.b - rsum) / hv[4];
            R = xn;
        } else {
            const int pq = act ? cur.p : 0;
            const int m = act ? s_mask[pq] : 0;
            double vv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) vv[q] = ((UM >> q) & 1u) ? s_val[pq * 9 + q] : 0.0;
            double rsum = 0.0;
            double t_;
            if ((UM >> 0) & 1u) { t_ = rsum + vv[0] * U0; rsum = ((m >> 0) & 1) ? t_ : rsum; }
            if ((UM >> 1) & 1u) { t_ = rsum + vv[1] * U1; rsum = ((m >> 1) & 1) ? t_ : rsum; }
            if ((UM >> 2) & 1u) { t_ = rsum + vv[2] * U2; rsum = ((m >> 2) & 1) ? t_ : rsum; }
            if ((UM >> 3) & 1u) { t_ = rsum + vv[3] * R; rsum = ((m >> 3) & 1) ? t_ : rsum; }
            if ((UM >> 5) & 1u) { t_ = rsum + vv[5] * O1; rsum = ((m >> 5) & 1) ? t_ : rsum; }
            if ((UM >> 6) & 1u) { t_ = rsum + vv[6] * D0; rsum = ((m >> 6) & 1) ? t_ : rsum; }
            if ((UM >> 7) & 1u) { t_ = rsum + vv[7] * D1; rsum = ((m >> 7) & 1) ? t_ : rsum; }
            if ((UM >> 8) & 1u) { t_ = rsum + vv[8] * D2; rsum = ((m >> 8) & 1) ? t_ : rsum; }
            const double diag = ((m >> 4) & 1) ? vv[4] : 0.0;
            const double q_ = (cur.b - rsum) / (diag != 0.0 ? diag : 1.0);
            xn = diag != 0.0 ? q_ : O0;
            R = act ? xn : R;
        }
        // the new value replaces the old one in the tile -- ONE STEP LATER (at the top of the next step), off the end of the step's
        // dependent chain.  Nothing reads the tile copy of a new value before its chunk is flushed (lane l + 1 gets it by DPP),
        // and the chunk that leaves at the top of a period ends 64 columns before the columns still pending.  No branch around
        // the write: an inactive lane writes a dump slot.
        // (Tried and dropped: a second, branch-free body for periods in which every element of chunks p - 4 .. p holds the
        // frequent pattern -- no id read, no activity tests, no wave-uniform branch: bit-exact, but 4.30 instead of 3.69 ms at
        // 4097^2 (twice the code for a lone wave to fetch; not investigated further).)
        w_addr = act ? a_x : w_dump;                                   // (= taddr(rowoff, frow, x): the address (y, x + 1) had a step ago)
        a_x = a_x1;
        w_val = xn;
      }
    }
    *reinterpret_cast<double *>(s_tile + w_addr) = w_val;
    // the chunks still in the ring leave, then everything of this band is done
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int fc = issued_flush + 1; fc <= cmax; ++fc) flush_chunk(fc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(W);
}

template <unsigned UM>
int launch_lds(GArgs a, hipStream_t st)
{
    if (a.sweeps > 1)
        hipLaunchKernelGGL((gs_band_lds_kernel<UM, true>), dim3((unsigned)(a.nbands * a.sweeps)), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((gs_band_lds_kernel<UM, false>), dim3((unsigned)a.nbands), dim3(64), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <unsigned UM>
int launch(GArgs a, hipStream_t st)
{
    if (a.sweeps > 1)
        hipLaunchKernelGGL((gs_wavefront_kernel<UM, true>), dim3((unsigned)(a.nbands * a.sweeps)), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((gs_wavefront_kernel<UM, false>), dim3((unsigned)a.nbands), dim3(64), 0, st, a);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int g_gs_max_sweeps = kMaxSweeps;           // sweeps pipelined in one launch (1 = a launch per sweep)
int g_gs_multi_max_rows = 8000000;          // ... on levels of at most this many rows
int g_gs_lds_min_rows = 0;                 // LDS bands instead of register bands from this many rows (gsw_lds = -1)
int g_gs_lds9 = 1;                          // LDS bands for 9-point operators too (32-line bands, two columns of skew per lane)
int g_gs_lds_multi = 1;                     // LDS bands: sweeps of a smoothing step pipelined in one launch (0: a launch per sweep)
int g_gs_lds = -1;                          // bands staged through LDS (gs_band_lds_kernel): -1 = wherever possible from g_gs_lds_min_rows rows, 0 = never, 1 = wherever possible

}  // namespace

int lmg_gsw_tune_set(const char *key, int v)
{
    if (strcmp(key, "gsw_max_sweeps") == 0) {
        if (v < 1 || v > kMaxSweeps) return LMG_ERR_ARG;
        g_gs_max_sweeps = v;
        return LMG_OK;
    }
    if (strcmp(key, "gsw_multi_max_rows") == 0) {
        if (v < 0) return LMG_ERR_ARG;
        g_gs_multi_max_rows = v;
        return LMG_OK;
    }
    if (strcmp(key, "gsw_lds") == 0) {
        if (v < -1 || v > 1) return LMG_ERR_ARG;
        g_gs_lds = v;
        return LMG_OK;
    }
    if (strcmp(key, "gsw_lds9") == 0) {
        if (v < 0 || v > 1) return LMG_ERR_ARG;
        g_gs_lds9 = v;
        return LMG_OK;
    }
    if (strcmp(key, "gsw_lds_multi") == 0) {
        if (v < 0 || v > 1) return LMG_ERR_ARG;
        g_gs_lds_multi = v;
        return LMG_OK;
    }
    return LMG_ERR_ARG;
}
int lmg_gsw_tune_get(const char *key)
{
    if (strcmp(key, "gsw_max_sweeps") == 0) return g_gs_max_sweeps;
    if (strcmp(key, "gsw_multi_max_rows") == 0) return g_gs_multi_max_rows;
    if (strcmp(key, "gsw_lds") == 0) return g_gs_lds;
    if (strcmp(key, "gsw_lds_multi") == 0) return g_gs_lds_multi;
    if (strcmp(key, "gsw_lds9") == 0) return g_gs_lds9;
    return LMG_ERR_ARG;
}

extern "C" {

int lmg_stencil_gs_supported(uint32_t union_mask)
{
    return union_mask == kMask5 || union_mask == kMask9 || union_mask == kMask7 || union_mask == kMask1D;
}

int64_t lmg_stencil_gs_work_bytes(int64_t n, int32_t line_stride)
{
    if (n <= 0 || line_stride <= 0) return 0;
    const int64_t lines = (n + line_stride - 1) / line_stride, nbands = (lines + 31) / 32;     // (the LDS bands of 9-point operators: 32 lines)
    return 4 * (kMaxSweeps * nbands + 3) + 4;
}

int lmg_stencil_gs_sweep(int64_t n, int32_t line_stride, const uint8_t *pid, int32_t npat, const double *st_val,
                         const int32_t *st_mask, uint32_t union_mask, int32_t hot_pattern, const double *h_hot_val,
                         double *x, const double *b, void *work, int sweeps, void *stream)
{
    if (n < 0 || n >= (1ll << 29) - 8192 || npat < 1 || npat > kMaxPat || sweeps < 0) return LMG_ERR_ARG;        // 32-bit byte offsets
    if (n == 0 || sweeps == 0) return LMG_OK;
    if (n < 2) return LMG_ERR_CAPACITY;
    if (!pid || !st_val || !st_mask || !x || !b || !work || line_stride < 3 || line_stride > n) return LMG_ERR_ARG;
    if (!lmg_stencil_gs_supported(union_mask)) return LMG_ERR_CAPACITY;
    GArgs a;
    a.n = (int)n;
    a.W = line_stride;
    a.lines = (int)((n + line_stride - 1) / line_stride);
    a.nbands = (a.lines + 63) / 64;
    a.npat = npat;
    a.pid = pid;
    a.st_val = st_val;
    a.st_mask = st_mask;
    a.x = x;
    a.b = b;
    a.work = reinterpret_cast<int *>(work);
    a.hot = -1;
    for (int k = 0; k < 9; ++k) a.hot_val[k] = 0.0;
    a.hot_rcp = 0.0;
    if (hot_pattern >= 0 && hot_pattern < npat && h_hot_val && h_hot_val[4] != 0.0) {
        a.hot = hot_pattern;
        for (int k = 0; k < 9; ++k) a.hot_val[k] = h_hot_val[k];
        int e = 0;
        const double mant = frexp(h_hot_val[4], &e);
        // +-2^k, with 1 / d still a normal number: t / d and t * (1 / d) are the same correctly rounded number
        if ((mant == 0.5 || mant == -0.5) && e > -1000 && e < 1000) a.hot_rcp = 1.0 / h_hot_val[4];
    }
    hipStream_t st = lmg_stream(stream);
    a.sweeps = 1;
    for (int sw = 0; sw < sweeps; sw += a.sweeps) {
        // several sweeps per launch only while the level lives in the Infinity Cache: their loads of x bypass L2 (16 bytes
        // per lane and iteration straight from memory), which costs more than the saved pipeline fills beyond it
        // (3 sweeps: 513^2 1.19 vs 2.43 ms, 1025^2 2.40 vs 4.80, 2049^2 7.2 vs 10.4, 3073^2 15.1 vs 16.0, 4097^2 26.8 vs 22.4)
        int per_launch = n <= g_gs_multi_max_rows ? g_gs_max_sweeps : 1;
        // bands staged through LDS: 5- / 7- / 9-point operators, 16-byte aligned vectors (the piece that straddles the end of a
        // vector of odd length is then inside its last 16 bytes), at least one full chunk per line
        const bool lds_ok = (union_mask == kMask5 || union_mask == kMask7 || (union_mask == kMask9 && g_gs_lds9)) && line_stride >= 64 &&
                            lmg_aligned16(x) && lmg_aligned16(b) && (reinterpret_cast<uintptr_t>(pid) & 3u) == 0;
        // (one sweep: 513^2 0.47 ms with LDS bands, 0.83 with register bands; three sweeps pipelined in one launch, LDS bands vs
        // register bands: 513^2 0.63 vs 1.21 ms, 1025^2 1.08 vs 2.58, 2049^2 1.97 vs 7.7, 4097^2 3.9 vs 22.6 -- the LDS bands take
        // every operator they can)
        const bool use_lds = lds_ok && g_gs_lds != 0 && (g_gs_lds == 1 || n >= g_gs_lds_min_rows);
        if (use_lds) per_launch = g_gs_lds_multi ? g_gs_max_sweeps : 1;       // (the LDS bands pipeline their sweeps at every size)
        a.sweeps = sweeps - sw < per_launch ? sweeps - sw : per_launch;
        // ticket and progress counters back to zero (a memset node when captured into a hipGraph)
        // (the error flag at [0] is cleared by the caller once and stays set)
        a.nbands = (use_lds && union_mask == kMask9) ? (a.lines + 31) / 32 : (a.lines + 63) / 64;
        if (hipMemsetAsync(a.work + 1, 0, 4 * (size_t)(a.sweeps * a.nbands + 2), st) != hipSuccess) return LMG_ERR_LAUNCH;
        int rc;
        if (use_lds) {
            rc = union_mask == kMask5 ? launch_lds<kMask5>(a, st) : union_mask == kMask7 ? launch_lds<kMask7>(a, st) : launch_lds<kMask9>(a, st);
            if (rc != LMG_OK) return rc;
            continue;
        }
        switch (union_mask) {
        case kMask5: rc = launch<kMask5>(a, st); break;
        case kMask9: rc = launch<kMask9>(a, st); break;
        case kMask7: rc = launch<kMask7>(a, st); break;
        default: rc = launch<kMask1D>(a, st); break;
        }
        if (rc != LMG_OK) return rc;
    }
    return LMG_OK;
}

}  // extern "C"
