// CSR transpose on the device (setup: R = P^T as an explicit CSR, Multigrid.py:93 evaluates `i.T @ res`).
//
// Counting sort by column instead of a general key sort: (1) histogram of the columns, (2) exclusive scan
// = row pointers of A^T, (3) every entry takes the next free slot of its column (atomic cursor: the order
// inside a column is arbitrary), (4) every row of A^T is sorted by its column index (= source row, unique
// within the row), which makes the result deterministic and equal to SciPy's `A.T.tocsr()` for a sorted,
// duplicate-free A.  Rows of transfer operators have a handful to a few dozen entries, so step (4) is an
// insertion sort per row; the caller falls back to a library sort for rows beyond kMaxSortRow.
#include "lmg_common.hpp"

namespace {

constexpr int kB = 256;
constexpr int kMaxSortRow = 512;

__global__ void __launch_bounds__(kB) col_count_kernel(int64_t nnz, const int *__restrict__ colidx, int *counts)
{
    for (int64_t e = (int64_t)blockIdx.x * kB + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kB)
        atomicAdd(&counts[colidx[e]], 1);
}

__global__ void __launch_bounds__(kB) transpose_fill_kernel(int64_t n, const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx,
                                                            const double *__restrict__ vals,
                                                            const int *__restrict__ t_rowptr, int *cursor,
                                                            int *t_colidx, double *t_vals)
{
    for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int c = colidx[e];
            const int pos = t_rowptr[c] + atomicAdd(&cursor[c], 1);
            t_colidx[pos] = (int)i;
            t_vals[pos] = vals[e];
        }
}

__global__ void __launch_bounds__(kB) sort_rows_kernel(int64_t m, const int *__restrict__ t_rowptr, int *t_colidx,
                                                       double *t_vals)
{
    for (int64_t r = (int64_t)blockIdx.x * kB + threadIdx.x; r < m; r += (int64_t)gridDim.x * kB) {
        const int s = t_rowptr[r], e = t_rowptr[r + 1];
        for (int a = s + 1; a < e; ++a) {
            const int kc = t_colidx[a];
            const double kv = t_vals[a];
            int b = a - 1;
            while (b >= s && t_colidx[b] > kc) {
                t_colidx[b + 1] = t_colidx[b];
                t_vals[b + 1] = t_vals[b];
                --b;
            }
            t_colidx[b + 1] = kc;
            t_vals[b + 1] = kv;
        }
    }
}

unsigned grid_of(int64_t n)
{
    int64_t g = (n + kB - 1) / kB;
    if (g > 8192) g = 8192;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

int lmg_csr_transpose_max_row(void) { return kMaxSortRow; }

int lmg_csr_transpose_count(int64_t nnz, int64_t ncols, const int32_t *colidx, int32_t *counts, void *stream)
{
    if (nnz < 0 || ncols < 0) return LMG_ERR_ARG;
    if (nnz == 0) return LMG_OK;
    if (!colidx || !counts) return LMG_ERR_ARG;
    col_count_kernel<<<grid_of(nnz), kB, 0, lmg_stream(stream)>>>(nnz, colidx, counts);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_csr_transpose_fill(int64_t n, int64_t ncols, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                           const int32_t *t_rowptr, int32_t *cursor, int32_t *t_colidx, double *t_vals, void *stream)
{
    if (n < 0 || ncols < 0) return LMG_ERR_ARG;
    if (n == 0 || ncols == 0) return LMG_OK;
    if (!rowptr || !t_rowptr || !cursor) return LMG_ERR_ARG;
    hipStream_t st = lmg_stream(stream);
    transpose_fill_kernel<<<grid_of(n), kB, 0, st>>>(n, rowptr, colidx, vals, t_rowptr, cursor, t_colidx, t_vals);
    sort_rows_kernel<<<grid_of(ncols), kB, 0, st>>>(ncols, t_rowptr, t_colidx, t_vals);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
