// Small dense fp64 helpers of the coarse-solver SETUP (coarse.py: block inverses by Schur recursion, Schur complement,
// W = A_II^-1 A_IS): a batched strided GEMM and a batched strided 2-D copy.  They replace rocBLAS / library elementwise
// kernels there for one reason only: the first use of each of those costs 50 - 200 ms of code-object loading in a fresh
// process (measured, tools/cold_ops.py), more than the whole warm setup of the cfg#4 hierarchy -- these live in the code
// object that is loaded anyway.  Setup arithmetic: FMA contraction is fine here (the inverses are verified by their defect).
#include "lmg_common.hpp"

namespace {

constexpr int kTM = 64, kTN = 64, kTK = 16;      // workgroup tile, 256 threads, 4 x 4 outputs per thread

// C[b] = alpha * A[b] (M x K) * B[b] (K x N) + beta * C[b], all row-major with leading dimensions and batch strides
__global__ void __launch_bounds__(256) gemm_kernel(int M, int N, int K, double alpha, const double *A, int64_t lda, int64_t sA,
                                                   const double *B, int64_t ldb, int64_t sB, double beta, double *C, int64_t ldc,
                                                   int64_t sC)
{
    __shared__ double As[kTK][kTM + 1];          // transposed: As[k][m]
    __shared__ double Bs[kTK][kTN + 1];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
    A += (int64_t)blockIdx.z * sA;
    B += (int64_t)blockIdx.z * sB;
    C += (int64_t)blockIdx.z * sC;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < K; k0 += kTK) {
        // A tile: 64 rows x 16 k (thread: row t / 4 .. , 4 consecutive k); B tile: 16 k x 64 columns
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;                 // 0 .. 1023
            const int ar = idx >> 4, ak = idx & 15;
            const int gm = m0 + ar, gk = k0 + ak;
            As[ak][ar] = (gm < M && gk < K) ? A[(int64_t)gm * lda + gk] : 0.0;
            const int bk = idx >> 6, bc = idx & 63;
            const int gk2 = k0 + bk, gn = n0 + bc;
            Bs[bk][bc] = (gk2 < K && gn < N) ? B[(int64_t)gk2 * ldb + gn] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kTK; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx + 16 * j;
            if (gn >= N) continue;
            double *c = C + (int64_t)gm * ldc + gn;
            *c = beta == 0.0 ? alpha * acc[i][j] : alpha * acc[i][j] + beta * *c;
        }
    }
}

// dst[b][r][c] = alpha * src[b][r][c] (+ dst when accumulate)
__global__ void __launch_bounds__(256) copy2d_kernel(int64_t rows, int64_t cols, double alpha, const double *src, int64_t lds_,
                                                     int64_t ss, double *dst, int64_t ldd, int64_t sd, int accumulate)
{
    src += (int64_t)blockIdx.z * ss;
    dst += (int64_t)blockIdx.z * sd;
    const int64_t total = rows * cols, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols, c = i - r * cols;
        const double v = alpha * src[r * lds_ + c];
        double *d = dst + r * ldd + c;
        *d = accumulate ? *d + v : v;
    }
}

}  // namespace

extern "C" {

int lmg_batched_gemm(int64_t batch, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     int64_t stride_a, const double *B, int64_t ldb, int64_t stride_b, double beta, double *C, int64_t ldc,
                     int64_t stride_c, void *stream)
{
    if (batch < 0 || M < 0 || N < 0 || K < 0 || batch > 65535 || M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31))
        return LMG_ERR_ARG;
    if (batch == 0 || M == 0 || N == 0) return LMG_OK;
    if (!C || (K > 0 && (!A || !B)) || lda < K || ldb < N || ldc < N) return LMG_ERR_ARG;
    const dim3 grid((unsigned)((N + kTN - 1) / kTN), (unsigned)((M + kTM - 1) / kTM), (unsigned)batch);
    if (grid.y > 65535) return LMG_ERR_CAPACITY;
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, lmg_stream(stream), (int)M, (int)N, (int)K, alpha, A, lda, stride_a, B, ldb,
                       stride_b, beta, C, ldc, stride_c);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

int lmg_copy2d(int64_t batch, int64_t rows, int64_t cols, double alpha, const double *src, int64_t ld_src, int64_t stride_src,
               double *dst, int64_t ld_dst, int64_t stride_dst, int accumulate, void *stream)
{
    if (batch < 0 || rows < 0 || cols < 0 || batch > 65535) return LMG_ERR_ARG;
    if (batch == 0 || rows == 0 || cols == 0) return LMG_OK;
    if (!src || !dst || ld_src < cols || ld_dst < cols) return LMG_ERR_ARG;
    int64_t g = (rows * cols + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)g, 1, (unsigned)batch), dim3(256), 0, lmg_stream(stream), rows, cols, alpha, src,
                       ld_src, stride_src, dst, ld_dst, stride_dst, accumulate);
    LMG_CHECK_LAUNCH();
    return LMG_OK;
}

}  // extern "C"
