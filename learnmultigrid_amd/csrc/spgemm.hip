// Galerkin product on gfx950: row-wise SpGEMM  C = A * B  (CSR x CSR -> CSR, sorted rows).
//
// Replaces the two csr_matmat passes behind `i.T @ A @ i` (Multigrid.py:97-98).  SpGEMM is
// integer-heavy and HBM/latency-bound; there is nothing MFMA-shaped in it.
//
// One workgroup owns one row of C at a time (grid-stride over rows):
//   expand   : thread l takes the l-th entry a_ij of the A row and streams B's row j,
//              writing (key, a_ij*b_jk) at its slot of an LDS product list; the slot
//              offsets come from a workgroup scan of the B row lengths, so the list is
//              in the exact order SciPy's Gustavson loop visits the products;
//   sort     : bitonic sort of the list in LDS by key = (column << 32 | sequence number)
//              -- the sequence number makes the sort stable;
//   compress : segment heads get their output slot from a scan of head flags, and each
//              head adds its segment IN ORDER, so c_ik is the same left-to-right sum of
//              separately rounded products that SciPy computes (bit-identical values);
//              unlike SciPy, entries that cancel to exactly 0.0 are kept (explicit zeros).
// The symbolic pass is the same pipeline on 32-bit keys only and returns nnz(C_i).
// Three size classes (64-thread workgroup with 128 or 1024 slots, 256-thread workgroup
// with 8192 slots) are chosen per launch from the largest product count of any row.
#include "lmg_common.hpp"

namespace {

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < LMG_WAVE; off <<= 1) {
        const int u = __shfl_up(v, off, LMG_WAVE);
        if (lane >= off) v += u;
    }
    return v;
}

// Exclusive scan of one int per thread across the workgroup; *total gets the sum.
// s_w holds BLOCK/64 + 1 ints.  Contains barriers: call from all threads.
template <int BLOCK>
__device__ __forceinline__ int block_excl_scan(int v, int *s_w, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan_i(v, lane);
    if (BLOCK == LMG_WAVE) {
        *total = __shfl(incl, 63, LMG_WAVE);
        return incl - v;
    }
    __syncthreads();                       // s_w may still be read from a previous call
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / LMG_WAVE; ++w) {
        const int sw = s_w[w];
        if (w < wave) woff += sw;
        tot += sw;
    }
    *total = tot;
    return woff + incl - v;
}

__device__ __forceinline__ int next_pow2(int v)
{
    int m = 1;
    while (m < v) m <<= 1;
    return m;
}

template <int BLOCK, typename KEY, bool WITH_VAL>
__device__ __forceinline__ void bitonic_sort_lds(KEY *key, double *val, int m)
{
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += BLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const KEY a = key[i], b = key[p];
                if ((a > b) == up) {
                    key[i] = b;
                    key[p] = a;
                    if (WITH_VAL) {
                        const double va = val[i], vb = val[p];
                        val[i] = vb;
                        val[p] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256) count_products_kernel(int64_t a_rows, const int *Ap,
                                                             const int *Aj, const int *Bp,
                                                             int *row_products, int *max_products)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int np = 0;
    if (i < a_rows) {
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            const int j = Aj[jj];
            np += Bp[j + 1] - Bp[j];
        }
        row_products[i] = np;
    }
    // one atomic per wave
    int m = np;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, LMG_WAVE));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(max_products, m);
}

// RECORD (numeric only): also store where every product went -- dst[prod_ptr[row] + sequence
// number] = its position in the sorted list -- and where the segment of every C entry ends, so
// that later numeric passes on the same pattern can skip the sort (spgemm_replay_kernel).
template <int BLOCK, int CAP, bool NUMERIC, bool RECORD = false>
__global__ void __launch_bounds__(BLOCK) spgemm_row_kernel(
    int64_t a_rows, const int *Ap, const int *Aj, const double *Ax, const int *Bp, const int *Bj,
    const double *Bx, const int *row_products, int *c_rownnz, const int *Cp, int *Cj, double *Cx,
    const int64_t *prod_ptr = nullptr, uint16_t *dst = nullptr, uint16_t *segend = nullptr)
{
    using KEY = typename std::conditional<NUMERIC, unsigned long long, unsigned int>::type;
    constexpr KEY kPad = ~(KEY)0;
    __shared__ KEY s_key[CAP];
    __shared__ double s_val[NUMERIC ? CAP : 1];
    __shared__ int s_w[BLOCK / LMG_WAVE + 1];

    for (int64_t row = blockIdx.x; row < a_rows; row += gridDim.x) {
        const int np = row_products[row];
        if (np == 0) {
            if (!NUMERIC && threadIdx.x == 0) c_rownnz[row] = 0;
            continue;
        }
        if (np > CAP) continue;   // long row: handled by spgemm_long_row_kernel
        const int a_s = Ap[row], a_e = Ap[row + 1];
        const int m = next_pow2(np);
        __syncthreads();          // previous row fully consumed
        // ---- expand -----------------------------------------------------------------
        int done = 0;
        for (int c0 = a_s; c0 < a_e; c0 += BLOCK) {
            const int jj = c0 + (int)threadIdx.x;
            int bs = 0, len = 0;
            double av = 0.0;
            if (jj < a_e) {
                const int j = Aj[jj];
                bs = Bp[j];
                len = Bp[j + 1] - bs;
                if (NUMERIC) av = Ax[jj];
            }
            int chunk_total;
            const int off = done + block_excl_scan<BLOCK>(len, s_w, &chunk_total);
            for (int kk = 0; kk < len; ++kk) {
                const int col = Bj[bs + kk];
                if (NUMERIC) {
                    s_key[off + kk] = (KEY)(((unsigned long long)(unsigned)col << 32) |
                                            (unsigned)(off + kk));
                    s_val[off + kk] = av * Bx[bs + kk];
                } else {
                    s_key[off + kk] = (KEY)(unsigned)col;
                }
            }
            done += chunk_total;
        }
        for (int p = np + (int)threadIdx.x; p < m; p += BLOCK) s_key[p] = kPad;
        __syncthreads();
        // ---- sort -------------------------------------------------------------------
        bitonic_sort_lds<BLOCK, KEY, NUMERIC>(s_key, s_val, m);
        if (RECORD) {
            const int64_t pb = prod_ptr[row];
            for (int p = threadIdx.x; p < np; p += BLOCK)
                dst[pb + (unsigned)((unsigned long long)s_key[p] & 0xFFFFFFFFull)] = (uint16_t)p;
        }
        // ---- compress ---------------------------------------------------------------
        int base_out = 0;
        for (int p0 = 0; p0 < np; p0 += BLOCK) {
            const int p = p0 + (int)threadIdx.x;
            bool head = false;
            unsigned col = 0;
            if (p < np) {
                col = NUMERIC ? (unsigned)((unsigned long long)s_key[p] >> 32) : (unsigned)s_key[p];
                if (p == 0) head = true;
                else {
                    const unsigned prev = NUMERIC ? (unsigned)((unsigned long long)s_key[p - 1] >> 32)
                                                  : (unsigned)s_key[p - 1];
                    head = prev != col;
                }
            }
            int nheads;
            const int rank = base_out + block_excl_scan<BLOCK>(head ? 1 : 0, s_w, &nheads);
            if (NUMERIC && head) {
                double sum = 0.0;
                int q = p;
                while (q < np && (unsigned)((unsigned long long)s_key[q] >> 32) == col) {
                    sum += s_val[q];
                    ++q;
                }
                const int o = Cp[row] + rank;
                Cj[o] = (int)col;
                Cx[o] = sum;
                if (RECORD) segend[o] = (uint16_t)q;
            }
            base_out += nheads;
        }
        if (!NUMERIC && threadIdx.x == 0) c_rownnz[row] = base_out;
    }
}

// ---- numeric pass on a recorded pattern ------------------------------------------------------
// Same expand step, but every product is written straight to its recorded sorted position; a C
// entry then is the in-order sum of one contiguous LDS range.  No sort, one barrier per row:
// the pass streams 2 B per product + the operands and is bound by HBM / L2, not by LDS sorting.
template <int BLOCK, int CAP>
__global__ void __launch_bounds__(BLOCK) spgemm_replay_kernel(
    int64_t a_rows, const int *Ap, const int *Aj, const double *Ax, const int *Bp, const double *Bx,
    const int *row_products, const int *Cp, double *Cx, const int64_t *prod_ptr, const uint16_t *dst,
    const uint16_t *segend)
{
    __shared__ double s_val[CAP];
    __shared__ int s_w[BLOCK / LMG_WAVE + 1];
    for (int64_t row = blockIdx.x; row < a_rows; row += gridDim.x) {
        const int np = row_products[row];
        if (np == 0 || np > CAP) continue;
        const int a_s = Ap[row], a_e = Ap[row + 1];
        const uint16_t *d = dst + prod_ptr[row];
        __syncthreads();          // previous row fully consumed
        int done = 0;
        for (int c0 = a_s; c0 < a_e; c0 += BLOCK) {
            const int jj = c0 + (int)threadIdx.x;
            int bs = 0, len = 0;
            double av = 0.0;
            if (jj < a_e) {
                const int j = Aj[jj];
                bs = Bp[j];
                len = Bp[j + 1] - bs;
                av = Ax[jj];
            }
            int chunk_total;
            const int off = done + block_excl_scan<BLOCK>(len, s_w, &chunk_total);
            for (int kk = 0; kk < len; ++kk) s_val[d[off + kk]] = av * Bx[bs + kk];
            done += chunk_total;
        }
        __syncthreads();
        const int c_s = Cp[row], c_n = Cp[row + 1] - c_s;
        for (int e = threadIdx.x; e < c_n; e += BLOCK) {
            int q = e ? (int)segend[c_s + e - 1] : 0;
            const int q_end = (int)segend[c_s + e];
            double sum = 0.0;
            for (; q < q_end; ++q) sum += s_val[q];
            Cx[c_s + e] = sum;
        }
    }
}

// ---- rows beyond the LDS capacity -----------------------------------------------------------
// One workgroup per long row with a DENSE accumulator (ncols(B) doubles + marks) in global
// scratch: the A entries of the row are visited one after the other (workgroup barrier in
// between) and the entries of each B row in parallel -- a canonical B row has distinct
// columns, so every c_ik still receives its products in SciPy's order.  The marks are then
// swept in column order to count (symbolic) or emit (numeric) the sorted row, and cleared.
// Cost O(products + ncols(B)) per row: meant for the few very long rows of an operator
// (dense-ish transfer columns), not for the bulk.
template <bool NUMERIC>
__global__ void __launch_bounds__(256) spgemm_long_row_kernel(
    int64_t nlong, const int *long_rows, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
    const int *Bj, const double *Bx, int64_t b_cols, double *scratch_val, int *scratch_mark,
    int *c_rownnz, const int *Cp, int *Cj, double *Cx)
{
    __shared__ int s_w[256 / LMG_WAVE + 1];
    double *acc = scratch_val + (int64_t)blockIdx.x * b_cols;
    int *mark = scratch_mark + (int64_t)blockIdx.x * b_cols;
    for (int64_t li = blockIdx.x; li < nlong; li += gridDim.x) {
        const int row = long_rows[li];
        for (int jj = Ap[row]; jj < Ap[row + 1]; ++jj) {
            const int j = Aj[jj];
            const double av = NUMERIC ? Ax[jj] : 0.0;
            for (int kk = Bp[j] + (int)threadIdx.x; kk < Bp[j + 1]; kk += 256) {
                const int col = Bj[kk];
                if (NUMERIC) acc[col] = (mark[col] ? acc[col] : 0.0) + av * Bx[kk];
                mark[col] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // sweep the marks in column order, 256 columns per step
        int out = NUMERIC ? Cp[row] : 0;
        for (int64_t c0 = 0; c0 < b_cols; c0 += 256) {
            const int64_t c = c0 + threadIdx.x;
            const int m = (c < b_cols) ? mark[c] : 0;
            int tot;
            const int rank = block_excl_scan<256>(m, s_w, &tot);
            if (m) {
                if (NUMERIC) {
                    Cj[out + rank] = (int)c;
                    Cx[out + rank] = acc[c];
                }
                mark[c] = 0;
            }
            out += tot;
        }
        if (!NUMERIC && threadIdx.x == 0) c_rownnz[row] = out;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

template <bool NUMERIC>
int launch_rows(int64_t a_rows, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                const int *Bj, const double *Bx, const int *row_products, int max_products,
                int *c_rownnz, const int *Cp, int *Cj, double *Cx, hipStream_t st)
{
    if (a_rows == 0) return LMG_OK;
    // rows above the LDS capacity are skipped here and done by lmg_spgemm_long_rows
    if (max_products > LMG_SPGEMM_MAX_ROW_PRODUCTS) max_products = LMG_SPGEMM_MAX_ROW_PRODUCTS;
    int64_t g = a_rows;
    if (max_products <= 128) {
        if (g > 256 * 32) g = 256 * 32;
        hipLaunchKernelGGL((spgemm_row_kernel<64, 128, NUMERIC>), dim3((unsigned)g), dim3(64), 0, st,
                           a_rows, Ap, Aj, Ax, Bp, Bj, Bx, row_products, c_rownnz, Cp, Cj, Cx);
    } else if (max_products <= 1024) {
        if (g > 256 * 10) g = 256 * 10;
        hipLaunchKernelGGL((spgemm_row_kernel<64, 1024, NUMERIC>), dim3((unsigned)g), dim3(64), 0, st,
                           a_rows, Ap, Aj, Ax, Bp, Bj, Bx, row_products, c_rownnz, Cp, Cj, Cx);
    } else {
        if (g > 256 * 2) g = 256 * 2;
        hipLaunchKernelGGL((spgemm_row_kernel<256, 8192, NUMERIC>), dim3((unsigned)g), dim3(256), 0, st,
                           a_rows, Ap, Aj, Ax, Bp, Bj, Bx, row_products, c_rownnz, Cp, Cj, Cx);
    }
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int launch_record(int64_t a_rows, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                  const int *Bj, const double *Bx, const int *row_products, int max_products,
                  const int *Cp, int *Cj, double *Cx, const int64_t *prod_ptr, uint16_t *dst,
                  uint16_t *segend, hipStream_t st)
{
    if (a_rows == 0) return LMG_OK;
    if (max_products > LMG_SPGEMM_MAX_ROW_PRODUCTS) max_products = LMG_SPGEMM_MAX_ROW_PRODUCTS;
    int64_t g = a_rows;
    if (max_products <= 128) {
        if (g > 256 * 32) g = 256 * 32;
        hipLaunchKernelGGL((spgemm_row_kernel<64, 128, true, true>), dim3((unsigned)g), dim3(64), 0, st, a_rows,
                           Ap, Aj, Ax, Bp, Bj, Bx, row_products, nullptr, Cp, Cj, Cx, prod_ptr, dst, segend);
    } else if (max_products <= 1024) {
        if (g > 256 * 10) g = 256 * 10;
        hipLaunchKernelGGL((spgemm_row_kernel<64, 1024, true, true>), dim3((unsigned)g), dim3(64), 0, st, a_rows,
                           Ap, Aj, Ax, Bp, Bj, Bx, row_products, nullptr, Cp, Cj, Cx, prod_ptr, dst, segend);
    } else {
        if (g > 256 * 2) g = 256 * 2;
        hipLaunchKernelGGL((spgemm_row_kernel<256, 8192, true, true>), dim3((unsigned)g), dim3(256), 0, st, a_rows,
                           Ap, Aj, Ax, Bp, Bj, Bx, row_products, nullptr, Cp, Cj, Cx, prod_ptr, dst, segend);
    }
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

template <int BLOCK, int CAP>
void launch_replay_class(int64_t a_rows, int per_cu, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                         const double *Bx, const int *row_products, const int *Cp, double *Cx,
                         const int64_t *prod_ptr, const uint16_t *dst, const uint16_t *segend, hipStream_t st)
{
    int64_t g = a_rows;
    if (g > 256 * (int64_t)per_cu) g = 256 * (int64_t)per_cu;
    hipLaunchKernelGGL((spgemm_replay_kernel<BLOCK, CAP>), dim3((unsigned)g), dim3(BLOCK), 0, st, a_rows, Ap, Aj,
                       Ax, Bp, Bx, row_products, Cp, Cx, prod_ptr, dst, segend);
}

}  // namespace

extern "C" {

int lmg_spgemm_numeric_record(int64_t a_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                              const int32_t *Bp, const int32_t *Bj, const double *Bx,
                              const int32_t *row_products, int32_t max_products, const int32_t *Cp,
                              int32_t *Cj, double *Cx, const int64_t *prod_ptr, uint16_t *dst,
                              uint16_t *segend, void *stream)
{
    if (a_rows < 0 || !Ap || !Bp || !row_products || !Cp || !prod_ptr || !dst || !segend) return LMG_ERR_ARG;
    return launch_record(a_rows, Ap, Aj, Ax, Bp, Bj, Bx, row_products, max_products, Cp, Cj, Cx, prod_ptr, dst,
                         segend, lmg_stream(stream));
}

int lmg_spgemm_numeric_replay(int64_t a_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                              const int32_t *Bp, const double *Bx, const int32_t *row_products,
                              int32_t max_products, const int32_t *Cp, double *Cx, const int64_t *prod_ptr,
                              const uint16_t *dst, const uint16_t *segend, void *stream)
{
    if (a_rows < 0 || !Ap || !Bp || !row_products || !Cp || !prod_ptr || !dst || !segend) return LMG_ERR_ARG;
    if (a_rows == 0) return LMG_OK;
    hipStream_t st = lmg_stream(stream);
    if (max_products > LMG_SPGEMM_MAX_ROW_PRODUCTS) max_products = LMG_SPGEMM_MAX_ROW_PRODUCTS;
    if (max_products <= 128)
        launch_replay_class<64, 128>(a_rows, 32, Ap, Aj, Ax, Bp, Bx, row_products, Cp, Cx, prod_ptr, dst, segend, st);
    else if (max_products <= 512)
        launch_replay_class<64, 512>(a_rows, 32, Ap, Aj, Ax, Bp, Bx, row_products, Cp, Cx, prod_ptr, dst, segend, st);
    else if (max_products <= 2048)
        launch_replay_class<128, 2048>(a_rows, 8, Ap, Aj, Ax, Bp, Bx, row_products, Cp, Cx, prod_ptr, dst, segend, st);
    else
        launch_replay_class<256, 8192>(a_rows, 2, Ap, Aj, Ax, Bp, Bx, row_products, Cp, Cx, prod_ptr, dst, segend, st);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_spgemm_count(int64_t a_rows, const int32_t *Ap, const int32_t *Aj, const int32_t *Bp,
                     int32_t *row_products, int32_t *max_products, void *stream)
{
    if (a_rows < 0 || !Ap || !Bp || !row_products || !max_products) return LMG_ERR_ARG;
    hipStream_t st = lmg_stream(stream);
    if (hipMemsetAsync(max_products, 0, sizeof(int32_t), st) != hipSuccess) return LMG_ERR_LAUNCH;
    if (a_rows == 0) return LMG_OK;
    const unsigned grid = (unsigned)((a_rows + 255) / 256);
    hipLaunchKernelGGL(count_products_kernel, dim3(grid), dim3(256), 0, st, a_rows, Ap, Aj, Bp,
                       row_products, max_products);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_spgemm_symbolic(int64_t a_rows, const int32_t *Ap, const int32_t *Aj, const int32_t *Bp,
                        const int32_t *Bj, const int32_t *row_products, int32_t max_products,
                        int32_t *c_rownnz, void *stream)
{
    if (a_rows < 0 || !Ap || !Bp || !row_products || !c_rownnz) return LMG_ERR_ARG;
    return launch_rows<false>(a_rows, Ap, Aj, nullptr, Bp, Bj, nullptr, row_products, max_products,
                              c_rownnz, nullptr, nullptr, nullptr, lmg_stream(stream));
}

int lmg_spgemm_numeric(int64_t a_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                       const int32_t *Bp, const int32_t *Bj, const double *Bx,
                       const int32_t *row_products, int32_t max_products, const int32_t *Cp,
                       int32_t *Cj, double *Cx, void *stream)
{
    if (a_rows < 0 || !Ap || !Bp || !row_products || !Cp) return LMG_ERR_ARG;
    return launch_rows<true>(a_rows, Ap, Aj, Ax, Bp, Bj, Bx, row_products, max_products, nullptr,
                             Cp, Cj, Cx, lmg_stream(stream));
}

int lmg_spgemm_long_rows(int numeric, int64_t nlong, const int32_t *long_rows, const int32_t *Ap,
                         const int32_t *Aj, const double *Ax, const int32_t *Bp, const int32_t *Bj,
                         const double *Bx, int64_t b_cols, int32_t nsets, double *scratch_val,
                         int32_t *scratch_mark, int32_t *c_rownnz, const int32_t *Cp, int32_t *Cj,
                         double *Cx, void *stream)
{
    if (nlong < 0 || b_cols < 0 || nsets < 1) return LMG_ERR_ARG;
    if (nlong == 0) return LMG_OK;
    if (!long_rows || !Ap || !Aj || !Bp || !Bj || !scratch_mark) return LMG_ERR_ARG;
    if (numeric && (!Ax || !Bx || !scratch_val || !Cp || !Cj || !Cx)) return LMG_ERR_ARG;
    if (!numeric && !c_rownnz) return LMG_ERR_ARG;
    const unsigned grid = (unsigned)(nlong < nsets ? nlong : nsets);
    if (numeric)
        hipLaunchKernelGGL(spgemm_long_row_kernel<true>, dim3(grid), dim3(256), 0, lmg_stream(stream), nlong,
                           long_rows, Ap, Aj, Ax, Bp, Bj, Bx, b_cols, scratch_val, scratch_mark, c_rownnz, Cp, Cj, Cx);
    else
        hipLaunchKernelGGL(spgemm_long_row_kernel<false>, dim3(grid), dim3(256), 0, lmg_stream(stream), nlong,
                           long_rows, Ap, Aj, Ax, Bp, Bj, Bx, b_cols, scratch_val, scratch_mark, c_rownnz, Cp, Cj, Cx);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
