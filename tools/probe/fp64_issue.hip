// Probe: issue rate and dependent latency of fp64 VALU ops for a single wave on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH>
__global__ void chains(double *out, int iters, double a, double b)
{
    double v[CH];
    for (int k = 0; k < CH; ++k) v[k] = threadIdx.x + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) v[k] = v[k] * a + b;      // contracted to v_fma_f64
    }
    double s = 0;
    for (int k = 0; k < CH; ++k) s += v[k];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
template <int CH>
__global__ void chains_muladd(double *out, int iters, double a, double b)
{
    double v[CH];
    for (int k = 0; k < CH; ++k) v[k] = threadIdx.x + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) v[k] = __dadd_rn(__dmul_rn(v[k], a), b);
    }
    double s = 0;
    for (int k = 0; k < CH; ++k) s += v[k];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
template <typename F>
float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *out; hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int blocks : {1, 1024}) {
        for (int threads : {64, 256, 512}) {
            float t1 = timeit([&] { chains<1><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
            float t4 = timeit([&] { chains<4><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
            float t8 = timeit([&] { chains<8><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
            float m4 = timeit([&] { chains_muladd<4><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
            printf("blocks %d threads %d: fma chain1 %.1f ns/op, 4 chains %.2f ns/op, 8 chains %.2f ns/op ; mul+add 4 chains %.2f ns/op-pair\n",
                   blocks, threads, t1 * 1e6 / (iters * 16.0), t4 * 1e6 / (iters * 16.0 * 4), t8 * 1e6 / (iters * 16.0 * 8), m4 * 1e6 / (iters * 16.0 * 4));
        }
    }
    return 0;
}
