// Probe: where one wave of the fused smoothing pass spends its cycles (per step: wait for the line, stage 1,
// stages 2..S + store, residual + store).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I learnmultigrid_amd/csrc \
//         -DLMG_FUSED_TRACE=<wave item> -o tools/probe/fused_trace.bin tools/probe/fused_trace.hip
#include "../../learnmultigrid_amd/csrc/stencil_fused.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int lmg_last_error_dummy;
int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 1025;
    const int seg = argc > 2 ? atoi(argv[2]) : 0;
    const int pf = argc > 3 ? atoi(argv[3]) : 2;
    const int resid = argc > 4 ? atoi(argv[4]) : 1;
    const long n = (long)side * side;
    std::vector<unsigned char> pid(n, 0);
    std::vector<double> x(n), b(n);
    for (long i = 0; i < n; ++i) { x[i] = (i * 37 % 101) * 0.01; b[i] = (i * 13 % 97) * 0.02; }
    double st_val[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};
    int st_mask = 0x0BA;
    unsigned char *dpid; double *dx, *db, *dout, *dr, *dval; int *dmask;
    if (hipMalloc(&dpid, n) != hipSuccess) return 1;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dout, n * 8); (void)hipMalloc(&dr, n * 8);
    (void)hipMalloc(&dval, 72); (void)hipMalloc(&dmask, 4);
    (void)hipMemcpy(dpid, pid.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dval, st_val, 72, hipMemcpyHostToDevice);
    (void)hipMemcpy(dmask, &st_mask, 4, hipMemcpyHostToDevice);
    lmg_fused_tune_set("fused_seg_lines", seg);
    lmg_fused_tune_set("fused_pf", pf);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        int rc = lmg_stencil_smooth(n, side, dpid, 1, dval, dmask, 0x0BA, 0, st_val, 3, dx, db, 0.8, dout, resid ? dr : nullptr, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d  %.4f ms\n", rc, ms);
    }
    unsigned long long tr[64 * 8];
    (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_fused_trace), sizeof tr);
    printf("whole loop of the traced wave: %llu ticks of the 100 MHz clock = %.2f us, %llu counts of s_memtime\n", tr[63 * 8 + 6],
           tr[63 * 8 + 6] * 0.01, tr[63 * 8 + 7]);
    printf("step: wait-line  stage1  stages2..S+store  resid+store  | step total (cycles)\n");
    for (int s = 0; s < 63 && tr[s * 8]; ++s) {
        unsigned long long *r = tr + s * 8;
        printf("%2d: %6llu %6llu %6llu %6llu | %6llu\n", s, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3],
               s + 1 < 64 && tr[(s + 1) * 8] ? tr[(s + 1) * 8] - r[0] : 0ull);
    }
    return 0;
}
