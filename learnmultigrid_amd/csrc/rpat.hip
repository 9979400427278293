// Row-pattern sweeps for gfx950: residual / Jacobi / SpMV on matrices whose rows repeat.
//
// An assembled grid operator (the 5-point Poisson matrix of config #4, its 9-point Galerkin
// coarsenings) consists of a handful of distinct rows when a row is written as
//     (length; column - row index, value bits of every entry, in storage order).
// RPAT stores each distinct row once (<= 255 patterns, <= 1024 entries in total, held in LDS)
// and one uint8 pattern id per row.  It is a lossless re-encoding of the CSR matrix -- built
// from the CSR arrays by the kernels at the end of this file, every row verified entry by
// entry against its pattern -- and the sweep walks the same entries in the same order with the
// same separately rounded products, so the results are bit-identical to sweep.hip / pcsr.hip.
// What changes is the byte count: 1 B/row of matrix instead of 16 (PCSR) or 64 (CSR) for
// the 5-point operator, i.e. the sweep moves x, b and out and almost nothing else.
// Matrices without repeating rows (variable coefficients, learned transfer operators,
// rectangular transfers) do not qualify and stay on PCSR / CSR.
//
// Kernel: persistent 256-thread workgroups, kRpt rows per thread (row = tile start + k*256 + t,
// so the gathers of a wave stay coalesced), XCD-aware tile ownership like pcsr.hip, pattern
// table in LDS (lanes of a wave almost always share a pattern: broadcast reads), JU entries of
// each of the kRpt rows per step => kRpt*JU independent gathers in flight per thread.
#include "lmg_common.hpp"

namespace {

enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2 };

constexpr int kBlock = 256;
constexpr int kMaxPat = 256;      // pattern ids are uint8; id 255 is never used by the builder
constexpr int kMaxEnt = 1024;     // total entries of all patterns (LDS: 12 KB)

// Column base of a row for RECTANGULAR grid operators (transfers between nested grids): with
// y = row / row_len, x = row % row_len,
//     base(row) = (y >> ysh) * col_stride + ((x >> xsh) << xshl),
// and a pattern stores  column - base(row).  Restriction R = P^T of the tensor-product interpolator
// (coarse row (Y, X) reads fine columns (2Y+dy) * Wf + 2X + dx): row_len = Wc, col_stride = 2 Wf,
// xshl = 1.  Prolongation P (fine row (y, x) reads coarse columns around (y >> 1, x >> 1)): row_len =
// Wf, col_stride = Wc, ysh = xsh = 1.  1-D transfers: row_len > number of rows (y = 0).
// row_len == 0 means base(row) = row (square operators).
struct GridMap {
    int row_len;
    int col_stride;
    int ysh, xsh, xshl;
    unsigned magic;             // floor(2^32 / row_len)
};

__device__ __forceinline__ int lmg_grid_base(const GridMap &m, int row)
{
    unsigned y = __umulhi((unsigned)row, m.magic);                  // row / row_len or one less (row < 2^31)
    unsigned x = (unsigned)row - y * (unsigned)m.row_len;
    if (x >= (unsigned)m.row_len) {
        x -= (unsigned)m.row_len;
        ++y;
    }
    return (int)((y >> m.ysh) * (unsigned)m.col_stride + ((x >> m.xsh) << m.xshl));
}

struct RArgs {
    GridMap map;
    int n;
    int tiles;
    int tiles_per_xcd;
    int npat;
    int nent;
    const int *pat_ptr;        // npat + 1
    const int *pat_off;        // nent: column - row
    const double *pat_val;     // nent
    const unsigned char *pid;  // n
    const double *x;
    const double *b;
    double *out;
    double alpha, beta;
    double *partial;
};

// NT: the streams without reuse (pattern ids, b, out) bypass the caches.  It pays when the vectors do
// not fit the 256 MB Infinity Cache anyway (cfg#4 fine level: -8 %) and costs 20 % when they do
// (2049^2), so the launcher turns it on from 8 M rows.
#ifndef LMG_RPAT_NT_MODE
#define LMG_RPAT_NT_MODE 3          // bit 0: nontemporal loads of ids / b, bit 1: nontemporal stores of out
#endif
template <int MODE, int JU, int kRpt, bool NT, bool MAP = false>
__global__ void __launch_bounds__(kBlock) rpat_sweep_kernel(RArgs a)
{
    constexpr bool NTL = NT && (LMG_RPAT_NT_MODE & 1), NTS = NT && (LMG_RPAT_NT_MODE & 2);
    constexpr int kTileRows = kBlock * kRpt;
    __shared__ int s_ptr[kMaxPat + 1];
    __shared__ int s_off[kMaxEnt];
    __shared__ double s_val[kMaxEnt];
    __shared__ double s_diag[MODE == MODE_JACOBI ? kMaxPat : 1];
    __shared__ double s_rdiag[MODE == MODE_JACOBI ? kMaxPat : 1];
    __shared__ double s_red[kBlock / LMG_WAVE];

    const int t = threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int t_begin = xcd * a.tiles_per_xcd;
    const int t_end = min(a.tiles, t_begin + a.tiles_per_xcd);
    if (t_begin + slot >= t_end) return;

    // Pattern ids and right-hand sides travel through registers two tiles ahead, in two register
    // sets (A, B) used alternately.  The loop body is straight-line on purpose: the loads of set A
    // are issued before the gathers of the tile that uses set B, so waiting for those gathers
    // (vmcnt counts in order) already guarantees set A without any wait at the top of the loop.
    // With a one-tile prefetch the loop top needed s_waitcnt vmcnt(0), which also waited for the
    // previous tile's STORE: two serialized memory round trips per tile instead of one.
    auto load_tile = [&](int tile, int (&pat)[kRpt], double (&bv)[kRpt]) {
        const int tl = tile < t_end ? tile : t_end - 1;          // past the end: harmless re-read
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            const int r = tl * kTileRows + k * kBlock + t;
            pat[k] = r < a.n ? (int)(NTL ? __builtin_nontemporal_load(a.pid + r) : a.pid[r]) : 0;
            bv[k] = (MODE != MODE_SPMV && r < a.n) ? (NTL ? __builtin_nontemporal_load(a.b + r) : a.b[r]) : 0.0;
        }
    };
    auto process = [&](int tile, const int (&pat)[kRpt], const double (&bv)[kRpt]) {
        const int r0 = tile * kTileRows;
        double local = 0.0;
        int row[kRpt], cb[kRpt], ps[kRpt], len[kRpt];
        double acc[kRpt], xi[kRpt];
        int maxlen = 0;
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            row[k] = r0 + k * kBlock + t;
            cb[k] = MAP ? lmg_grid_base(a.map, row[k] < a.n ? row[k] : 0) : row[k];
            ps[k] = s_ptr[pat[k]];
            len[k] = row[k] < a.n ? s_ptr[pat[k] + 1] - ps[k] : 0;
            maxlen = max(maxlen, len[k]);
            acc[k] = 0.0;
            xi[k] = 0.0;
        }
        for (int j0 = 0; j0 < maxlen; j0 += JU) {
            int off[JU][kRpt];
            double v[JU][kRpt], xv[JU][kRpt];
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) {
                    const bool act = j0 + jj < len[k];
                    const int p = act ? ps[k] + j0 + jj : 0;
                    off[jj][k] = s_off[p];
                    v[jj][k] = s_val[p];
                }
            }
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) {
                    const bool act = j0 + jj < len[k];
                    xv[jj][k] = a.x[act ? cb[k] + off[jj][k] : 0];
                }
            }
#pragma unroll
            for (int jj = 0; jj < JU; ++jj) {
#pragma unroll
                for (int k = 0; k < kRpt; ++k) {
                    const bool act = j0 + jj < len[k];
                    const double s2 = acc[k] + v[jj][k] * xv[jj][k];
                    acc[k] = act ? s2 : acc[k];
                    if (MODE == MODE_JACOBI) xi[k] = (act && off[jj][k] == 0) ? xv[jj][k] : xi[k];
                }
            }
        }
        // Free on the usual path (the last gather was just waited for with vmcnt(0)); it tells the
        // compiler's wait-count tracking that the other register set is complete on EVERY path,
        // including the one that skips the gather loop (all rows of the tile empty).
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int k = 0; k < kRpt; ++k) {
            if (row[k] < a.n) {
                if (MODE == MODE_RESIDUAL) {
                    const double r = bv[k] - acc[k];
                    if (a.out) {
                        if (NTS) __builtin_nontemporal_store(r, a.out + row[k]);
                        else a.out[row[k]] = r;
                    }
                    local += r * r;
                } else if (MODE == MODE_JACOBI) {
                    const double r = bv[k] - acc[k];
                    if (s_diag[pat[k]] != 0.0) {
                        const double o = xi[k] + a.alpha * (s_rdiag[pat[k]] * r);
                        if (NTS) __builtin_nontemporal_store(o, a.out + row[k]);
                        else a.out[row[k]] = o;
                    }
                    else a.out[row[k]] = a.x[row[k]];
                } else {
                    double s = acc[k];
                    if (a.alpha != 1.0) s = a.alpha * s;
                    if (a.beta == 0.0) {
                        if (NTS) __builtin_nontemporal_store(s, a.out + row[k]);
                        else a.out[row[k]] = s;
                    } else if (a.beta == 1.0) a.out[row[k]] = a.out[row[k]] + s;
                    else a.out[row[k]] = a.beta * a.out[row[k]] + s;
                }
            }
        }
        if (MODE == MODE_RESIDUAL && a.partial != nullptr) {
            __syncthreads();
            const double tot = lmg_block_sum<kBlock>(local, s_red);
            if (t == 0) a.partial[tile] = tot;
        }
    };
    int patA[kRpt], patB[kRpt];
    double bA[kRpt], bB[kRpt];
    int tile = t_begin + slot;
    load_tile(tile, patA, bA);
    load_tile(tile + nslots, patB, bB);
    // the pattern tables are staged while the first two tiles' ids and right-hand sides are in flight: on the small
    // levels a workgroup has one tile, and table -> ids -> gathers would be three dependent trips to memory
    for (int i = t; i <= a.npat; i += kBlock) s_ptr[i] = a.pat_ptr[i];
    for (int i = t; i < a.nent; i += kBlock) {
        s_off[i] = a.pat_off[i];
        s_val[i] = a.pat_val[i];
    }
    __syncthreads();
    if (MODE == MODE_JACOBI) {
        // diagonal of every pattern: its entries with offset 0, summed in storage order like
        // the CSR sweep does per row; one division per pattern and workgroup instead of per row
        for (int p = t; p < a.npat; p += kBlock) {
            double d = 0.0;
            for (int j = s_ptr[p]; j < s_ptr[p + 1]; ++j)
                if (s_off[j] == 0) d += s_val[j];
            s_diag[p] = d;
            s_rdiag[p] = d != 0.0 ? 1.0 / d : 0.0;
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the loop is entered with both sets complete
    while (tile + nslots < t_end) {
        process(tile, patA, bA);
        load_tile(tile + 2 * nslots, patA, bA);
        process(tile + nslots, patB, bB);
        load_tile(tile + 3 * nslots, patB, bB);
        tile += 2 * nslots;
    }
    if (tile < t_end) process(tile, patA, bA);
}

__global__ void __launch_bounds__(1024) rpat_reduce_partials_kernel(const double *partial, int64_t count, double *out)
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

int g_rpat_variant = 0;      // 0 = pick from the longest pattern; 1..4 = forced (tuning)

template <int MODE, int JU, int kRpt, bool NT, bool MAP = false>
int launch_nt(RArgs a, hipStream_t st)
{
    a.tiles = (a.n + kBlock * kRpt - 1) / (kBlock * kRpt);
    a.tiles_per_xcd = (a.tiles + 7) / 8;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rpat_sweep_kernel<MODE, JU, kRpt, NT, MAP>, kBlock, 0) !=
            hipSuccess || per_cu < 1)
        per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    int64_t grid = 256 * (int64_t)per_cu;
    if (grid > (int64_t)a.tiles_per_xcd * 8) grid = (int64_t)a.tiles_per_xcd * 8;
    hipLaunchKernelGGL((rpat_sweep_kernel<MODE, JU, kRpt, NT, MAP>), dim3((unsigned)grid), dim3(kBlock), 0, st, a);
    LMG_CHECK_LAUNCH();
    return a.tiles;
}

int g_rpat_nt_rows = 1 << 23;      // rows from which the id / b / out streams bypass the caches (tunable)

template <int MODE, int JU, int kRpt>
int launch_one(RArgs a, hipStream_t st)
{
    if (MODE == MODE_SPMV && a.map.row_len > 0)       // rectangular grid operators (transfers): SpMV only
        return a.n >= g_rpat_nt_rows ? launch_nt<MODE_SPMV, JU, kRpt, true, true>(a, st)
                                     : launch_nt<MODE_SPMV, JU, kRpt, false, true>(a, st);
    return a.n >= g_rpat_nt_rows ? launch_nt<MODE, JU, kRpt, true>(a, st) : launch_nt<MODE, JU, kRpt, false>(a, st);
}

template <int MODE>
int launch(RArgs a, int maxlen, hipStream_t st)
{
    int v = g_rpat_variant;
    // measured on MI355X (tools/time_rpat.py): all geometries are within 10 % of each other
    // (prolongations of nested grids -- rows of 1, 2 or 4 entries, read-modify-write of the fine vector --
    // want more rows in flight per lane: 4097^2 <- 2049^2 0.083 ms with geometry 3 vs 0.090-0.100)
    if (v == 0) v = (a.map.row_len > 0 && maxlen <= 4) ? 3 : (maxlen <= 5 ? 2 : 4);
    switch (v) {
    case 1: return launch_one<MODE, 5, 1>(a, st);
    case 2: return launch_one<MODE, 5, 2>(a, st);
    case 3: return launch_one<MODE, 1, 4>(a, st);
    case 4: return launch_one<MODE, 3, 2>(a, st);
    default: return LMG_ERR_ARG;
    }
}

// ---- building the format (setup) -------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// 64-bit hash of (length; column - row, value bits ...) of every row; never the all-ones value
__global__ void __launch_bounds__(256) row_pattern_hash_kernel(int64_t n, GridMap map, const int *__restrict__ rowptr,
                                                               const int *__restrict__ colidx,
                                                               const unsigned long long *__restrict__ vbits,
                                                               unsigned long long *__restrict__ hash)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = rowptr[i], e = rowptr[i + 1];
    const int base = map.row_len > 0 ? lmg_grid_base(map, (int)i) : (int)i;
    unsigned long long h = mix64(0x9E3779B97F4A7C15ull + (unsigned long long)(e - s));
    for (int j = s; j < e; ++j) {
        h = mix64(h ^ (unsigned long long)(unsigned)(colidx[j] - base));
        h = mix64(h ^ vbits[j]);
    }
    if (h == ~0ull) h = 0;
    hash[i] = h;
}

// one representative row per pattern: whoever gets there first (every row is verified later)
__global__ void __launch_bounds__(256) pattern_claim_kernel(int64_t n, const unsigned char *__restrict__ pid, int *rep)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int p = i < n ? (int)pid[i] : -1;
    // neighbouring rows mostly share their pattern: only the first lane of every run asks
    const int prev = __shfl_up(p, 1, LMG_WAVE);
    if (p < 0 || ((threadIdx.x & (LMG_WAVE - 1)) != 0 && prev == p)) return;
    if (__atomic_load_n(&rep[p], __ATOMIC_RELAXED) < 0) atomicCAS(&rep[p], -1, (int)i);
}

// every row against its pattern, entry by entry (hash collisions must not pass)
__global__ void __launch_bounds__(256) pattern_verify_kernel(int64_t n, int64_t ncols, GridMap map,
                                                             const int *__restrict__ rowptr,
                                                             const int *__restrict__ colidx,
                                                             const unsigned long long *__restrict__ vbits,
                                                             const unsigned char *__restrict__ pid, int npat,
                                                             const int *__restrict__ pat_ptr,
                                                             const int *__restrict__ pat_off,
                                                             const unsigned long long *__restrict__ pat_vbits,
                                                             int *mismatch)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = pid[i];
    bool ok = p < npat;
    if (ok) {
        const int s = rowptr[i], e = rowptr[i + 1], ps = pat_ptr[p];
        const int base = map.row_len > 0 ? lmg_grid_base(map, (int)i) : (int)i;
        ok = (e - s) == pat_ptr[p + 1] - ps;
        for (int j = 0; ok && j < e - s; ++j) {
            const int c = colidx[s + j];
            ok = (c - base == pat_off[ps + j]) && (vbits[s + j] == pat_vbits[ps + j]) && c >= 0 && c < ncols;
        }
    }
    if (!ok) *mismatch = 1;
}

}  // namespace

int lmg_rpat_tune_set(int v)
{
    if (v >= 1000) { g_rpat_nt_rows = v; return LMG_OK; }      // >= 1000: nontemporal threshold in rows (legacy spelling)
    if (v < 0 || v > 4) return LMG_ERR_ARG;
    g_rpat_variant = v;
    return LMG_OK;
}
int lmg_rpat_tune_get(void) { return g_rpat_variant; }
// "rpat_nt_rows": rows from which the streams without reuse bypass the caches (1 = always, the parity
// tests force the nontemporal instantiations on small matrices this way)
int lmg_rpat_nt_set(int v)
{
    if (v < 1) return LMG_ERR_ARG;
    g_rpat_nt_rows = v;
    return LMG_OK;
}
int lmg_rpat_nt_get(void) { return g_rpat_nt_rows; }

extern "C" {

int lmg_rpat_limits(int32_t *max_patterns, int32_t *max_entries)
{
    if (max_patterns) *max_patterns = kMaxPat - 1;
    if (max_entries) *max_entries = kMaxEnt;
    return LMG_OK;
}

// h_grid_map: NULL / row_len 0 = square operator, else {row_len, col_stride, ysh, xsh, xshl}
static int make_grid_map(const int32_t *h, GridMap *m)
{
    m->row_len = 0;
    m->col_stride = 0;
    m->ysh = m->xsh = m->xshl = 0;
    m->magic = 0;
    if (!h || h[0] == 0) return LMG_OK;
    if (h[0] < 1 || h[1] < 0 || h[2] < 0 || h[2] > 1 || h[3] < 0 || h[3] > 1 || h[4] < 0 || h[4] > 1) return LMG_ERR_ARG;
    m->row_len = h[0];
    m->col_stride = h[1];
    m->ysh = h[2];
    m->xsh = h[3];
    m->xshl = h[4];
    m->magic = h[0] == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned long long)h[0]);
    return LMG_OK;
}

int lmg_rpat_sweep(int mode, int64_t n, const uint8_t *pid, int32_t npat, int32_t nent, int32_t max_len,
                   const int32_t *pat_ptr, const int32_t *pat_off, const double *pat_val, const double *x,
                   const double *b, double *out, double alpha, double beta, double *partials, double *norm2,
                   void *stream)
{
    return lmg_rpat_sweep_grid(mode, n, nullptr, pid, npat, nent, max_len, pat_ptr, pat_off, pat_val, x, b, out,
                               alpha, beta, partials, norm2, stream);
}

int lmg_rpat_sweep_grid(int mode, int64_t n, const int32_t *h_grid_map, const uint8_t *pid, int32_t npat,
                        int32_t nent, int32_t max_len, const int32_t *pat_ptr, const int32_t *pat_off,
                        const double *pat_val, const double *x, const double *b, double *out, double alpha,
                        double beta, double *partials, double *norm2, void *stream)
{
    GridMap gm;
    if (make_grid_map(h_grid_map, &gm) != LMG_OK) return LMG_ERR_ARG;
    if (gm.row_len > 0 && mode != MODE_SPMV) return LMG_ERR_ARG;         // rectangular operators: SpMV only
    if (n < 0 || n >= INT32_MAX || npat < 1 || npat >= kMaxPat || nent < 0 || nent > kMaxEnt) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!pid || !pat_ptr || !x || (nent > 0 && (!pat_off || !pat_val))) return LMG_ERR_ARG;
    if (mode == MODE_SPMV) {
        if (!out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_JACOBI) {
        if (!b || !out || x == out) return LMG_ERR_ARG;
    } else if (mode == MODE_RESIDUAL) {
        if (!b || (partials == nullptr) != (norm2 == nullptr) || (!out && !partials)) return LMG_ERR_ARG;
    } else {
        return LMG_ERR_ARG;
    }
    RArgs a;
    a.map = gm;
    a.n = (int)n;
    a.tiles = a.tiles_per_xcd = 0;
    a.npat = npat;
    a.nent = nent;
    a.pat_ptr = pat_ptr;
    a.pat_off = pat_off;
    a.pat_val = pat_val;
    a.pid = pid;
    a.x = x;
    a.b = b;
    a.out = out;
    a.alpha = alpha;
    a.beta = beta;
    a.partial = (mode == MODE_RESIDUAL) ? partials : nullptr;
    hipStream_t st = lmg_stream(stream);
    int tiles;
    if (mode == MODE_RESIDUAL) tiles = launch<MODE_RESIDUAL>(a, max_len, st);
    else if (mode == MODE_JACOBI) tiles = launch<MODE_JACOBI>(a, max_len, st);
    else tiles = launch<MODE_SPMV>(a, max_len, st);
    if (tiles < 0) return tiles;
    if (mode == MODE_RESIDUAL && partials) {
        hipLaunchKernelGGL(rpat_reduce_partials_kernel, dim3(1), dim3(1024), 0, st, partials, (int64_t)tiles, norm2);
        LMG_CHECK_LAUNCH();
    }
    return LMG_OK;
}

int lmg_rpat_row_hash(int64_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, uint64_t *hash,
                      void *stream)
{
    return lmg_rpat_row_hash_grid(n, nullptr, rowptr, colidx, vals, hash, stream);
}

int lmg_rpat_row_hash_grid(int64_t n, const int32_t *h_grid_map, const int32_t *rowptr, const int32_t *colidx,
                           const double *vals, uint64_t *hash, void *stream)
{
    GridMap gm;
    if (n < 0 || n >= INT32_MAX || make_grid_map(h_grid_map, &gm) != LMG_OK) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!rowptr || !hash) return LMG_ERR_ARG;
    row_pattern_hash_kernel<<<(unsigned)((n + 255) / 256), 256, 0, lmg_stream(stream)>>>(
        n, gm, rowptr, colidx, reinterpret_cast<const unsigned long long *>(vals),
        reinterpret_cast<unsigned long long *>(hash));
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_rpat_claim(int64_t n, const uint8_t *pid, int32_t *rep, void *stream)
{
    if (n < 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!pid || !rep) return LMG_ERR_ARG;
    pattern_claim_kernel<<<(unsigned)((n + 255) / 256), 256, 0, lmg_stream(stream)>>>(n, pid, rep);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_rpat_verify(int64_t n, int64_t ncols, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                    const uint8_t *pid, int32_t npat, const int32_t *pat_ptr, const int32_t *pat_off,
                    const double *pat_val, int32_t *mismatch, void *stream)
{
    return lmg_rpat_verify_grid(n, ncols, nullptr, rowptr, colidx, vals, pid, npat, pat_ptr, pat_off, pat_val,
                                mismatch, stream);
}

int lmg_rpat_verify_grid(int64_t n, int64_t ncols, const int32_t *h_grid_map, const int32_t *rowptr,
                         const int32_t *colidx, const double *vals, const uint8_t *pid, int32_t npat,
                         const int32_t *pat_ptr, const int32_t *pat_off, const double *pat_val, int32_t *mismatch,
                         void *stream)
{
    GridMap gm;
    if (n < 0 || n >= INT32_MAX || npat < 1 || make_grid_map(h_grid_map, &gm) != LMG_OK) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!rowptr || !pid || !pat_ptr || !mismatch) return LMG_ERR_ARG;
    pattern_verify_kernel<<<(unsigned)((n + 255) / 256), 256, 0, lmg_stream(stream)>>>(
        n, ncols, gm, rowptr, colidx, reinterpret_cast<const unsigned long long *>(vals), pid, npat, pat_ptr, pat_off,
        reinterpret_cast<const unsigned long long *>(pat_val), mismatch);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
