// valu_probe.hip -- issue rate of v_fma_f32 vs v_pk_fma_f32 (and the GELU forms built from them) on one MI355X:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
// Every thread runs a long chain of independent FMA streams; the kernel is pure VALU (no memory), so time / op count = issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    if (MODE == 0) {            // 8 scalar FMA chains
        for (int i = 0; i < iters; ++i) {      // asm: hipcc's SLP vectoriser pairs plain fmaf chains into v_pk_fma_f32 by itself
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    } else {                    // 4 packed FMA chains = the same 8 results per iteration
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        const f2 mm = {m, m}, cc = {c, c};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
        }
        a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

int main() {
    float* out;
    const int grid = 256 * 8, iters = 20000;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<grid, 256>>>(out, iters, 1.0f); else probe<1><<<grid, 256>>>(out, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)grid * 256 * iters * 8;
            printf("%s: %.3f ms  %.1f TFLOP/s (2 flop per FMA result)\n", mode ? "v_pk_fma_f32 x4" : "v_fma_f32 x8  ", ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
