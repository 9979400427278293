// Setup-side kernels: build the packed twin (PCSR, see lmg.h) of a CSR matrix on the device
// and the inverse diagonal used by the zero-initial-guess Jacobi sweep.  None of this is in
// the V-cycle itself (learn_multigrid/solvers/Multigrid.py:36-124 has no counterpart: SciPy
// keeps plain CSR); it is the format conversion done once per hierarchy, and once more per
// Galerkin rebuild for the value streams.  Everything here is one streaming pass over the
// matrix, HBM-bound.
#include "lmg_common.hpp"
#include <limits.h>

namespace {

constexpr int kPackBlock = 256;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;   // a NaN payload; tracked apart

// ---- column range of every tile, and the uint16 tile-relative columns ----------------
__global__ __launch_bounds__(kPackBlock) void tile_colrange_kernel(int64_t n, int tile_rows, int64_t ntile,
                                                                   const int32_t *__restrict__ rowptr,
                                                                   const int32_t *__restrict__ colidx,
                                                                   int32_t *__restrict__ cmin,
                                                                   int32_t *__restrict__ cmax)
{
    __shared__ int s_mn[kPackBlock / LMG_WAVE], s_mx[kPackBlock / LMG_WAVE];
    const int lane = threadIdx.x & (LMG_WAVE - 1), wave = threadIdx.x / LMG_WAVE;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t r0 = tile * tile_rows, r1 = (r0 + tile_rows < n) ? r0 + tile_rows : n;
        const int lo = rowptr[r0], hi = rowptr[r1];
        int mn = INT_MAX, mx = 0;
        for (int e = lo + threadIdx.x; e < hi; e += kPackBlock) {
            const int c = colidx[e];
            mn = c < mn ? c : mn;
            mx = c > mx ? c : mx;
        }
#pragma unroll
        for (int off = LMG_WAVE / 2; off > 0; off >>= 1) {
            const int a = __shfl_down(mn, off, LMG_WAVE), b = __shfl_down(mx, off, LMG_WAVE);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kPackBlock / LMG_WAVE; ++w) {
                mn = s_mn[w] < mn ? s_mn[w] : mn;
                mx = s_mx[w] > mx ? s_mx[w] : mx;
            }
            cmin[tile] = hi > lo ? mn : 0;
            cmax[tile] = hi > lo ? mx : 0;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kPackBlock) void encode_cols16_kernel(int64_t n, int tile_rows, int64_t ntile,
                                                                   const int32_t *__restrict__ rowptr,
                                                                   const int32_t *__restrict__ colidx,
                                                                   const int32_t *__restrict__ colbase,
                                                                   uint16_t *__restrict__ out)
{
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t r0 = tile * tile_rows, r1 = (r0 + tile_rows < n) ? r0 + tile_rows : n;
        const int lo = rowptr[r0], hi = rowptr[r1], base = colbase[tile];
        for (int e = lo + threadIdx.x; e < hi; e += kPackBlock) out[e] = (uint16_t)(colidx[e] - base);
    }
}

// ---- set of distinct values (bit patterns) -------------------------------------------
// Open-addressing table in global memory, linear probing, insert-only.  state[0] = distinct
// keys so far, state[1] = overflow (more than `limit` keys, or the table is too full to be
// trusted), state[2] = the all-ones bit pattern (the table's empty marker) occurs.  A small
// per-workgroup LDS filter absorbs the repeats: stencil matrices have a handful of values,
// and without it every lane of the chip would poll the same few table slots.
__device__ __forceinline__ uint64_t mix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

constexpr int kFilterSlots = 1024;
constexpr int kMaxProbe = 4096;

__global__ __launch_bounds__(kPackBlock) void value_set_insert_kernel(int64_t count,
                                                                      const unsigned long long *__restrict__ bits,
                                                                      unsigned long long *table, uint64_t mask,
                                                                      int limit, int *state)
{
    __shared__ unsigned long long s_seen[kFilterSlots];
    for (int i = threadIdx.x; i < kFilterSlots; i += kPackBlock) s_seen[i] = kEmpty;
    __syncthreads();
    volatile int *vstate = state;
    const int64_t stride = (int64_t)gridDim.x * kPackBlock;
    unsigned long long last = kEmpty;
    for (int64_t i = (int64_t)blockIdx.x * kPackBlock + threadIdx.x; i < count; i += stride) {
        const unsigned long long key = bits[i];
        if (key == kEmpty) { vstate[2] = 1; continue; }
        if (key == last) continue;       // already inserted by this thread
        last = key;
        const uint64_t h0 = mix64(key);
        const int f = (int)(h0 >> 40) & (kFilterSlots - 1);
        if (s_seen[f] == key) continue;
        if (vstate[1]) return;
        uint64_t h = h0 & mask;
        int probe = 0;
        for (; probe < kMaxProbe; ++probe) {
            const unsigned long long cur = __atomic_load_n(&table[h], __ATOMIC_RELAXED);
            if (cur == key) break;
            if (cur == kEmpty) {
                const unsigned long long prev = atomicCAS(&table[h], kEmpty, key);
                if (prev == kEmpty) {
                    if (atomicAdd(state, 1) + 1 > limit) vstate[1] = 1;
                    break;
                }
                if (prev == key) break;
            }
            h = (h + 1) & mask;
        }
        if (probe == kMaxProbe) { vstate[1] = 1; return; }
        s_seen[f] = key;                 // 64-bit LDS store: never torn; races only lose a hint
    }
}

// index of every value in the sorted dictionary (sorted as SIGNED 64-bit patterns)
constexpr int kDictLds = 2048;

template <typename OUT>
__global__ __launch_bounds__(kPackBlock) void value_encode_kernel(int64_t count, const long long *__restrict__ bits,
                                                                  const long long *__restrict__ dict, int ndict,
                                                                  OUT *__restrict__ out, int *missing)
{
    __shared__ long long s_dict[kDictLds];
    const bool in_lds = ndict <= kDictLds;
    if (in_lds) {
        for (int i = threadIdx.x; i < ndict; i += kPackBlock) s_dict[i] = dict[i];
        __syncthreads();
    }
    const int64_t stride = (int64_t)gridDim.x * kPackBlock;
    for (int64_t i = (int64_t)blockIdx.x * kPackBlock + threadIdx.x; i < count; i += stride) {
        const long long key = bits[i];
        int lo = 0, hi = ndict;          // lower bound
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const long long d = in_lds ? s_dict[mid] : dict[mid];
            if (d < key) lo = mid + 1; else hi = mid;
        }
        const long long found = lo < ndict ? (in_lds ? s_dict[lo] : dict[lo]) : ~key;
        if (found != key) *missing = 1;
        out[i] = (OUT)lo;
    }
}

// ---- 1 / a_ii ------------------------------------------------------------------------
__global__ __launch_bounds__(kPackBlock) void inverse_diagonal_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                                      const int32_t *__restrict__ colidx,
                                                                      const double *__restrict__ vals,
                                                                      double *__restrict__ dinv)
{
    const int64_t i = (int64_t)blockIdx.x * kPackBlock + threadIdx.x;
    if (i >= n) return;
    double d = 0.0;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
        if (colidx[e] == i) d += vals[e];
    dinv[i] = d != 0.0 ? 1.0 / d : 0.0;
}

inline int grid_for(int64_t work_items, int per_block, int cap)
{
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

// the occupied slots of the table, appended in any order to `out` (at most `cap`); d_count: how many there were
__global__ void __launch_bounds__(256) value_set_collect_kernel(const unsigned long long *table, long long slots,
                                                                       unsigned long long *out, int cap, int *count)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < slots; i += stride) {
        const unsigned long long k = table[i];
        if (k != ~0ull) {
            const int at = atomicAdd(count, 1);
            if (at < cap) out[at] = k;
        }
    }
}

extern "C" {

int lmg_pcsr_tile_colrange(int64_t n, int32_t tile_rows, const int32_t *d_rowptr, const int32_t *d_colidx,
                           int32_t *d_cmin, int32_t *d_cmax, void *stream)
{
    if (n < 0 || tile_rows <= 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!d_rowptr || !d_colidx || !d_cmin || !d_cmax) return LMG_ERR_ARG;
    const int64_t ntile = (n + tile_rows - 1) / tile_rows;
    tile_colrange_kernel<<<grid_for(ntile, 1, 1 << 20), kPackBlock, 0, lmg_stream(stream)>>>(
        n, tile_rows, ntile, d_rowptr, d_colidx, d_cmin, d_cmax);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_pcsr_encode_cols16(int64_t n, int32_t tile_rows, const int32_t *d_rowptr, const int32_t *d_colidx,
                           const int32_t *d_colbase, uint16_t *d_out, void *stream)
{
    if (n < 0 || tile_rows <= 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!d_rowptr || !d_colidx || !d_colbase || !d_out) return LMG_ERR_ARG;
    const int64_t ntile = (n + tile_rows - 1) / tile_rows;
    encode_cols16_kernel<<<grid_for(ntile, 1, 1 << 20), kPackBlock, 0, lmg_stream(stream)>>>(
        n, tile_rows, ntile, d_rowptr, d_colidx, d_colbase, d_out);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_value_set_collect(const uint64_t *d_table, int64_t table_slots, uint64_t *d_out, int32_t cap, int32_t *d_count,
                          void *stream)
{
    if (table_slots < 1 || cap < 0 || !d_table || !d_out || !d_count) return LMG_ERR_ARG;
    int64_t grid = (table_slots + 255) / 256;
    if (grid > 2048) grid = 2048;
    value_set_collect_kernel<<<(unsigned)grid, 256, 0, lmg_stream(stream)>>>(
        reinterpret_cast<const unsigned long long *>(d_table), (long long)table_slots,
        reinterpret_cast<unsigned long long *>(d_out), cap, d_count);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_value_set_insert(int64_t count, const double *d_vals, uint64_t *d_table, int64_t table_slots,
                         int32_t limit, int32_t *d_state, void *stream)
{
    if (count < 0 || limit < 0) return LMG_ERR_ARG;
    if (table_slots < 2 || (table_slots & (table_slots - 1))) return LMG_ERR_ARG;
    const int grid = grid_for(count, kPackBlock * 8, 1024);
    // every in-flight lane may add one key after the overflow flag went up
    if ((int64_t)limit + (int64_t)grid * kPackBlock > table_slots / 2) return LMG_ERR_ARG;
    if (count == 0) return LMG_OK;
    if (!d_vals || !d_table || !d_state) return LMG_ERR_ARG;
    value_set_insert_kernel<<<grid, kPackBlock, 0, lmg_stream(stream)>>>(
        count, reinterpret_cast<const unsigned long long *>(d_vals),
        reinterpret_cast<unsigned long long *>(d_table), (uint64_t)(table_slots - 1), limit, d_state);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_value_encode(int64_t count, const double *d_vals, const double *d_dict, int32_t ndict, int width,
                     void *d_out, int32_t *d_missing, void *stream)
{
    if (count < 0 || ndict < 1 || (width != 1 && width != 2)) return LMG_ERR_ARG;
    if ((width == 1 && ndict > 256) || ndict > 65536) return LMG_ERR_ARG;
    if (count == 0) return LMG_OK;
    if (!d_vals || !d_dict || !d_out || !d_missing) return LMG_ERR_ARG;
    const int grid = grid_for(count, kPackBlock * 8, 4096);
    const long long *bits = reinterpret_cast<const long long *>(d_vals);
    const long long *dict = reinterpret_cast<const long long *>(d_dict);
    if (width == 1)
        value_encode_kernel<uint8_t><<<grid, kPackBlock, 0, lmg_stream(stream)>>>(
            count, bits, dict, ndict, static_cast<uint8_t *>(d_out), d_missing);
    else
        value_encode_kernel<uint16_t><<<grid, kPackBlock, 0, lmg_stream(stream)>>>(
            count, bits, dict, ndict, static_cast<uint16_t *>(d_out), d_missing);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_csr_inverse_diagonal(int64_t n, const int32_t *d_rowptr, const int32_t *d_colidx, const double *d_vals,
                             double *d_dinv, void *stream)
{
    if (n < 0) return LMG_ERR_ARG;
    if (n == 0) return LMG_OK;
    if (!d_rowptr || !d_colidx || !d_vals || !d_dinv) return LMG_ERR_ARG;
    inverse_diagonal_kernel<<<(unsigned)((n + kPackBlock - 1) / kPackBlock), kPackBlock, 0, lmg_stream(stream)>>>(
        n, d_rowptr, d_colidx, d_vals, d_dinv);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
