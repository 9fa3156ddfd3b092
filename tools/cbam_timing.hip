// Phase timing of the single-read CBAM kernel (development aid, not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DCBAM_TIMING tools/cbam_timing.hip \
//         pytorch-attention_amd/csrc/cbam_single.hip pytorch-attention_amd/csrc/api.hip -o tools/bin/cbam_timing
// Prints, per phase, the mean time (us) between consecutive stamps over all workgroups and their first 16 slices.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pytorch-attention_amd/csrc/common.h"
namespace mi355 {
extern unsigned long long* g_cbam_dbg;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, C = 256, Cr = 16, H = 56, W = 56, ks = 7;
    const size_t n = (size_t)B * C * H * W;
    float *x, *y, *w1, *w2, *wc;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w1, Cr * C * 4); hipMalloc(&w2, C * Cr * 4); hipMalloc(&wc, 2 * ks * ks * 4);
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w1, h.data(), Cr * C * 4, hipMemcpyHostToDevice); hipMemcpy(w2, h.data() + 5000, C * Cr * 4, hipMemcpyHostToDevice);
    hipMemcpy(wc, h.data() + 9000, 2 * ks * ks * 4, hipMemcpyHostToDevice);
    const size_t extra = mi355::cbam_single_extra_bytes(B, C, H, W);
    void* ws; hipMalloc(&ws, extra);
    const int G = 256 * 2;
    unsigned long long* dbg; hipMalloc(&dbg, (size_t)G * 16 * 10 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipMemset(dbg, 0, (size_t)G * 16 * 10 * 8);
        mi355::g_cbam_dbg = (it == 3) ? dbg : nullptr;
        hipEventRecord(e0, 0);
        int rc = mi355::cbam_single(x, w1, w2, wc, y, B, C, Cr, H, W, ks, ws, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("run %d rc %d %.3f ms (memset included)\n", it, rc, ms);
    }
    std::vector<unsigned long long> d((size_t)G * 16 * 10);
    hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"loads+publish1", "hop1 wait", "reduce+hop2 wait", "gates MLP", "stats+publish3", "hop3 wait", "conv", "store issue"};
    double sum[9] = {0}; long cnt = 0; double cyc = 0; long ccnt = 0;
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < 16; ++k) {
            unsigned long long* p = &d[((size_t)g * 16 + k) * 10];
            if (!p[8]) continue;
            for (int i = 0; i < 8; ++i) sum[i] += (double)(p[i + 1] - p[i]);
            ++cnt;
            if (k + 1 < 16 && p[10]) { cyc += (double)(p[10] - p[0]); ++ccnt; }
        }
    printf("slices sampled %ld (100 MHz clock)\n", cnt);
    for (int i = 0; i < 8; ++i) printf("  %-18s %7.2f us\n", names[i], sum[i] / cnt / 100.0);
    printf("  slice-to-slice     %7.2f us\n", cyc / ccnt / 100.0);
    return 0;
}
