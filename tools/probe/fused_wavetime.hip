// Probe: start / end time of every wave of one fused smoothing pass (3 sweeps + residual) on a 5-point grid operator
// with identity boundary rows -- do the waves of the boundary strips / segments (per-lane pattern path) finish later
// than the interior ones (pattern in scalar registers)?  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I learnmultigrid_amd/csrc \
//         -DLMG_FUSED_WAVETIME=16384 -o tools/probe/fused_wavetime.bin tools/probe/fused_wavetime.hip
#include "../../learnmultigrid_amd/csrc/stencil_fused.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 4097;
    const int seg = argc > 2 ? atoi(argv[2]) : 0;
    const int resid = argc > 3 ? atoi(argv[3]) : 1;
    const int allhot = argc > 4 ? atoi(argv[4]) : 0;
    const long n = (long)side * side;
    std::vector<unsigned char> pid(n, 0);
    if (!allhot)
        for (int y = 0; y < side; ++y)
            for (int x = 0; x < side; ++x)
                if (y == 0 || x == 0 || y == side - 1 || x == side - 1) pid[(long)y * side + x] = 1;
    std::vector<double> x(n), b(n);
    for (long i = 0; i < n; ++i) { x[i] = (i * 37 % 101) * 0.01; b[i] = (i * 13 % 97) * 0.02; }
    double st_val[18] = {0, -1, 0, -1, 4, -1, 0, -1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
    int st_mask[2] = {0x0BA, 0x010};
    unsigned char *dpid; double *dx, *db, *dout, *dr, *dval; int *dmask;
    if (hipMalloc(&dpid, n) != hipSuccess) return 1;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dout, n * 8); (void)hipMalloc(&dr, n * 8);
    (void)hipMalloc(&dval, sizeof st_val); (void)hipMalloc(&dmask, sizeof st_mask);
    (void)hipMemcpy(dpid, pid.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dval, st_val, sizeof st_val, hipMemcpyHostToDevice);
    (void)hipMemcpy(dmask, st_mask, sizeof st_mask, hipMemcpyHostToDevice);
    lmg_fused_tune_set("fused_seg_lines", seg);
#ifdef LMG_FUSED_TRACE
    { int item = argc > 5 ? atoi(argv[5]) : 17 * 100000 + 2000; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_trace_item), &item, sizeof item); }
#endif
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        int rc = lmg_stencil_smooth(n, side, dpid, 2, dval, dmask, 0x0BA, 0, st_val, 3, dx, db, 0.8, dout, resid ? dr : nullptr, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d  %.4f ms\n", rc, ms);
    }
    static unsigned long long wt[2 * LMG_FUSED_WAVETIME];
    (void)hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_fused_wavetime), sizeof wt);
    int items = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < LMG_FUSED_WAVETIME; ++i)
        if (wt[2 * i + 1]) { items = i + 1; t0 = std::min(t0, wt[2 * i]); t1 = std::max(t1, wt[2 * i + 1]); }
    printf("side %d items %d; kernel span %.2f us\n", side, items, (t1 - t0) * 0.01);
    // classes (bit 0 of the end stamp): general body (boundary segments) / FAST body; the first items are the boundary strips
    const char *names[2] = {"general body", "FAST body"};
    for (int c = 0; c < 2; ++c) {
        std::vector<double> dur, end, start;
        for (int i = 0; i < items; ++i) {
            const int cls = (int)(wt[2 * i + 1] & 1ull);
            if (cls != c) continue;
            dur.push_back((wt[2 * i + 1] - wt[2 * i]) * 0.01);
            end.push_back((wt[2 * i + 1] - t0) * 0.01);
            start.push_back((wt[2 * i] - t0) * 0.01);
        }
        if (dur.empty()) continue;
        std::sort(dur.begin(), dur.end()); std::sort(end.begin(), end.end()); std::sort(start.begin(), start.end());
        const size_t m = dur.size();
        printf("%-17s %5zu waves: duration min %.1f med %.1f p90 %.1f max %.1f us | start med %.1f max %.1f | end med %.1f p90 %.1f max %.1f\n",
               names[c], m, dur[0], dur[m / 2], dur[m * 9 / 10], dur[m - 1], start[m / 2], start[m - 1], end[m / 2], end[m * 9 / 10], end[m - 1]);
    }
#ifdef LMG_FUSED_TRACE
    {
        unsigned long long tr[64 * 8];
        (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_fused_trace), sizeof tr);
        printf("traced wave, cycles per step: wait+arrive | issue loads | stage1 | stage2 | stage3 | resid etc. | step total\n");
        for (int k = 0; k < 40 && tr[k * 8 + 7]; ++k) {
            unsigned long long *r = tr + k * 8;
            printf("%2d: %6llu %6llu %6llu %6llu %6llu %6llu | %6llu\n", k, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4],
                   r[7] - r[5], k + 1 < 64 && tr[(k + 1) * 8] ? tr[(k + 1) * 8] - r[0] : 0ull);
        }
    }
#endif
    // histogram of wave end times (10 bins)
    int hist[10] = {0};
    for (int i = 0; i < items; ++i) hist[std::min(9, (int)((wt[2 * i + 1] - t0) * 10 / (t1 - t0 + 1)))]++;
    printf("end-time histogram (tenths of the span):");
    for (int k = 0; k < 10; ++k) printf(" %d", hist[k]);
    printf("\n");
    return 0;
}
