#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* x, double* out, int n)
{
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 8, 0x00020000);
    const int t = threadIdx.x;
    // lane 0: pair (n-1, n): second half out of range; lane 1: pair (-1, 0): offset wraps; lane 2: (n-2,n-1) valid; lane 3: (n, n+1)
    const long long i = t == 0 ? n - 1 : (t == 1 ? -1 : (t == 2 ? n - 2 : n));
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(i * 8), 0, 0);
    out[2 * t + 0] = __hiloint2double((int)v.y, (int)v.x);
    out[2 * t + 1] = __hiloint2double((int)v.w, (int)v.z);
}
__global__ void ks(double* x, int n)
{
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 8, 0x00020000);
    const int t = threadIdx.x;
    const long long i = t == 0 ? n - 1 : -1;
    u4 v;
    v.x = 0; v.y = 0x40590000; v.z = 0; v.w = 0x40690000;   // 100.0, 200.0
    if (t < 2) __builtin_amdgcn_raw_buffer_store_b128(v, rx, (unsigned)(i * 8), 0, 0);
}
int main()
{
    const int n = 100;
    double h[n + 8], *dx, *dout, ho[16];
    for (int i = 0; i < n + 8; ++i) h[i] = i + 1;
    hipMalloc(&dx, (n + 8) * 8); hipMalloc(&dout, 16 * 8);
    hipMemcpy(dx, h, (n + 8) * 8, hipMemcpyHostToDevice);
    k<<<1, 4>>>(dx, dout, n);
    hipMemcpy(ho, dout, 8 * 8, hipMemcpyDeviceToHost);
    printf("load (n-1,n): [%g %g]  (-1,0): [%g %g]  (n-2,n-1): [%g %g]  (n,n+1): [%g %g]\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7]);
    ks<<<1, 2>>>(dx, n);
    hipMemcpy(h, dx, (n + 8) * 8, hipMemcpyDeviceToHost);
    printf("after stores: x[0]=%g x[1]=%g x[n-2]=%g x[n-1]=%g x[n]=%g x[n+1]=%g\n", h[0], h[1], h[n - 2], h[n - 1], h[n], h[n + 1]);
    return 0;
}
