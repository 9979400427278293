// Probe: what a small dependent kernel costs inside a hipGraph on gfx950 -- empty, one memory round trip, two
// dependent round trips -- for grids of 136 and 2048 workgroups (the 257^2 and 1025^2 levels of the cycle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__global__ void k_empty(const double *, double *, const int *, long) {}
__global__ void k_one_trip(const double *x, double *y, const int *, long n)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] + 1.0;
}
__global__ void k_two_trips(const double *x, double *y, const int *idx, long n)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[idx[i]] + 1.0;
}
__global__ void k_lds_then_trip(const double *x, double *y, const int *idx, long n)
{
    __shared__ int tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = idx[threadIdx.x];
    __syncthreads();
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i + tab[threadIdx.x & 63] - tab[threadIdx.x & 63]] + 1.0;
}
typedef void (*kern_t)(const double *, double *, const int *, long);
int main()
{
    const long nmax = 2048 * 256;
    double *a, *b; int *idx;
    CK(hipMalloc(&a, nmax * 8)); CK(hipMalloc(&b, nmax * 8)); CK(hipMalloc(&idx, nmax * 4));
    CK(hipMemset(a, 0, nmax * 8)); CK(hipMemset(b, 0, nmax * 8));
    int *h = (int *)malloc(nmax * 4);
    for (long i = 0; i < nmax; ++i) h[i] = (int)i;
    CK(hipMemcpy(idx, h, nmax * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    const char *names[] = {"empty", "one trip (y = x + 1)", "two dependent trips (y = x[idx])", "table -> LDS -> sync -> trip"};
    kern_t ks[] = {k_empty, k_one_trip, k_two_trips, k_lds_then_trip};
    const int chain = 40;
    for (int wgs : {136, 2048}) {
        for (int k = 0; k < 4; ++k) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int c = 0; c < chain; ++c) {
                const double *src = (c & 1) ? b : a; double *dst = (c & 1) ? a : b;
                hipLaunchKernelGGL(ks[k], dim3(wgs), dim3(256), 0, st, src, dst, idx, (long)wgs * 256);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e0, st));
            const int reps = 20;
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%4d workgroups, %-36s %.2f us per kernel\n", wgs, names[k], ms * 1e3 / (reps * chain));
        }
    }
    return 0;
}
