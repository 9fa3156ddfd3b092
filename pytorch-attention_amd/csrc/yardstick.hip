// yardstick.hip -- box calibration for bench.py ("re-confirm with a measured stream-copy and a measured MFMA microbench").
//
// mi355_mfma_yardstick: register-operand MFMA loop, nothing but the matrix pipe -- what this box's silicon sustains under dense MFMA
// load (TFLOP/s) and at which shader clock (two independent readings: the wave's own cycle counter against the 100 MHz wall clock, and
// the 32x32x16 issue rate, which is exactly 32 cycles per SIMD back to back: MI355X_MICROARCH.md "Per-instruction cycle constants").
// A GEMM line of the bench that is low by the same factor as this number says "box"; one that is low while this is not says "kernel".
#include "common.h"
#include "mma.h"

namespace {
typedef float f16v __attribute__((ext_vector_type(16)));

// non-zero, lane-dependent operand bits: zero-filled operands clock ~19 % higher than random data on this part (DVFS give-back), so a
// yardstick fed with zeros would overstate what a real GEMM can reach
__device__ __forceinline__ h8 frag(unsigned seed) {
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        seed = seed * 1664525u + 1013904223u;
        v[e] = (_Float16)((float)((seed >> 9) & 1023u) * (1.0f / 512.0f) - 1.0f);          // [-1, 1)
    }
    return v;
}

// SHAPE 0: v_mfma_f32_16x16x32_f16, a 32 x 64 wave tile (2 A x 4 B fragments, 8 accumulators), the engine's instruction
// SHAPE 1: v_mfma_f32_32x32x16_f16, a 64 x 64 wave tile (2 A x 2 B fragments, 4 accumulators of 16 registers)
// Both: 131 072 FLOP per wave and iteration.  Two 256-thread workgroups per CU = two waves per SIMD.
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_yardstick_kernel(int iters, float* __restrict__ sink, unsigned long long* __restrict__ report) {
    const unsigned lane = threadIdx.x, blk = blockIdx.x;
    h8 a0 = frag(lane * 2654435761u + blk), a1 = frag(lane * 40503u + 77u + blk);
    h8 b0 = frag(lane * 69069u + 1u), b1 = frag(lane * 1103515245u + 12345u), b2 = frag(lane * 22695477u + 5u), b3 = frag(lane * 134775813u + 9u);
    unsigned long long c0 = 0, r0 = 0;
    const bool stamp = blk == 0 && threadIdx.x == 0;
    if (stamp) {
        c0 = __builtin_readcyclecounter();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    float out = 0.f;
    if constexpr (SHAPE == 0) {
        f4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
        // accumulators updated IN PLACE by inline assembly: left to the register allocator the eight chains came out rotated through
        // overlapping register ranges (D = a[24:27] from C = a[22:25], ...) and the loop ran at 25 cycles per MFMA instead of 17 --
        // round 5's first bench line read 1.09 PFLOP/s for this shape where the same silicon does 2.0 (tools/mfma_probe.hip)
        for (int it = 0; it < iters; ++it) {
#define Y16(J_, A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[J_]) : "v"(A_), "v"(B_))
            Y16(0, a0, b0); Y16(1, a0, b1); Y16(2, a0, b2); Y16(3, a0, b3);
            Y16(4, a1, b0); Y16(5, a1, b1); Y16(6, a1, b2); Y16(7, a1, b3);
#undef Y16
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    } else {
        f16v acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#define Y32(J_, A_, B_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[J_]) : "v"(A_), "v"(B_))
            Y32(0, a0, b0); Y32(1, a0, b1); Y32(2, a1, b0); Y32(3, a1, b1);
#undef Y32
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) out += acc[j][e];
        out += (float)b2[0] + (float)b3[0];
    }
    if (stamp) {
        report[0] = __builtin_readcyclecounter() - c0;                 // shader-clock ticks of this wave's whole loop
        report[1] = __builtin_amdgcn_s_memrealtime() - r0;             // the same interval on the constant 100 MHz counter
        report[2] = gridDim.x;                                         // workgroups launched (2 per CU), 4 waves each
    }
    if (out == 12345.678f) sink[0] = out;            // never true in practice: keeps the accumulators observable
}
}  // namespace

extern "C" int mi355_mfma_yardstick(int shape, int iters, float* sink, unsigned long long* report, mi355_stream_t stream) {
    MI355_CHECK_ARG(sink != nullptr && report != nullptr);
    MI355_CHECK_ARG(shape == 0 || shape == 1);
    MI355_CHECK_ARG(iters > 0 && iters <= (1 << 24));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // report (device, 3 x u64), written by workgroup 0: {shader-clock ticks, 100 MHz ticks, workgroups}.  MFMA instructions of the launch =
    // workgroups x 4 waves x iters x (8 | 4); FLOP per instruction 16 384 | 32 768
    const int grid = mi355::resident_slots(2);
    if (shape == 0) mfma_yardstick_kernel<0><<<grid, 256, 0, st>>>(iters, sink, report);
    else mfma_yardstick_kernel<1><<<grid, 256, 0, st>>>(iters, sink, report);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}
