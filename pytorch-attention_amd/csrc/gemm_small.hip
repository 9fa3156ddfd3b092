// gemm_small.hip -- nn.Linear on SMALL outputs (the classifier head of a ViT: ViT.py:191, M = 256 images, N = 1000 classes, K = 768).
//
// The fp32-in engine (gemm.hip) cuts C into 128 x 128 tiles: 16 workgroups for the ViT head -- 240 of the 256 CUs idle, 47-97 us for a
// 0.39 GFLOP product ("What's missing" 6).  Here a tile is 16 x 32 and belongs to ONE WAVE (a 64-thread workgroup):
// 512 workgroups for the head, two per CU.  No LDS at all: a lane loads the 8 consecutive fp32 k-values of its fragment row straight from
// global memory (two float4; X rows may be strided, e.g. token 0 of every image in place), converts them to the MFMA operand format in
// registers -- the rounding point of the engine -- and the K steps of a chunk are in flight together.  A row's K steps are accumulated
// in the same order with the same instruction as in gemm_kernel (one accumulator per output, 32 k per step, hi/lo triple in strict
// mode), so the result is BIT-IDENTICAL to the 128 x 128 kernel's (tests/test_round5_gpu.py).  Bias only (no activation / LayerScale /
// residual: those shapes stay on the engine).
#include "common.h"
#include "mma.h"

namespace {

struct SmallArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K, lda, ldb, ldc, tiles_n;
};

template <int PREC, int CH>
__global__ __launch_bounds__(64) void gemm_small_kernel(const SmallArgs g) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    constexpr int NS = M_::NSPLIT;
    const int lane = threadIdx.x, frow = lane & 15, fk = (lane >> 4) * 8;
    const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
    const int m0 = tm * 16, n0 = tn * 32;
    // fragment rows of this lane (clamped: rows past the edge load a valid row and are never stored)
    const int ma = m0 + frow;
    const float* ap = g.A + (long)(ma < g.M ? ma : g.M - 1) * g.lda + fk;
    const float* bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + frow;
        bp[j] = g.B + (long)(n < g.N ? n : g.N - 1) * g.ldb + fk;
    }
    f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};

    auto pack = [](f4 lo4, f4 hi4, v8* out) {                      // 8 fp32 -> operand fragment (hi, and lo in strict mode)
        const v4 a = M_::cvt(lo4), b = M_::cvt(hi4);
        out[0] = v8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if constexpr (NS == 2) {
            const v4 c = M_::cvt_lo(lo4, a), d = M_::cvt_lo(hi4, b);
            out[1] = v8{c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        }
    };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    struct Chunk { f4 a[CH][2], b[CH][2][2]; };                    // [k step]([B fragment])[half of the 8 k-values]
    auto load = [&](Chunk& c, int k0) {                            // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int ko = k0 + s * 32, k = ko + fk;
            c.a[s][0] = k < g.K ? *reinterpret_cast<const f4*>(ap + ko) : zero;
            c.a[s][1] = k + 4 < g.K ? *reinterpret_cast<const f4*>(ap + ko + 4) : zero;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                c.b[s][j][0] = k < g.K ? *reinterpret_cast<const f4*>(bp[j] + ko) : zero;
                c.b[s][j][1] = k + 4 < g.K ? *reinterpret_cast<const f4*>(bp[j] + ko + 4) : zero;
            }
        }
    };
    auto compute = [&](const Chunk& c, int k0) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (k0 + s * 32 >= g.K) break;                          // whole step past K (uniform): nothing to add
            v8 fa[NS], fb[2][NS];
            pack(c.a[s][0], c.a[s][1], fa);
#pragma unroll
            for (int j = 0; j < 2; ++j) pack(c.b[s][j][0], c.b[s][j][1], fb[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mma_step<PREC>(fa, fb[j], acc[j]);
        }
    };
    // two register sets: the loads of chunk c + 1 are in flight while chunk c is converted and multiplied (a lone wave per tile has nobody
    // else to hide its load latency behind)
    Chunk c0, c1;
    constexpr int KC = 32 * CH;
    load(c0, 0);
    for (int k0 = 0; k0 < g.K; k0 += 2 * KC) {
        if (k0 + KC < g.K) load(c1, k0 + KC);
        compute(c0, k0);
        if (k0 + 2 * KC < g.K) load(c0, k0 + 2 * KC);
        if (k0 + KC < g.K) compute(c1, k0 + KC);
    }
    // C / D layout: lane holds column (lane & 15), rows (lane >> 4) * 4 + [0, 4) of each 16 x 16 tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + frow;
        if (n >= g.N) continue;
        const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (lane >> 4) * 4 + r;
            if (m < g.M) g.C[(long)m * g.ldc + n] = g.bias ? acc[j][r] + bn : acc[j][r];
        }
    }
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the product is small enough that the 128 x 128 engine would leave most CUs idle.
int gemm_small_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc, int precision,
                  hipStream_t st) {
    if (!opt_gemm_small()) return MI355_EUNSUPPORTED;
    if ((K & 3) || (lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B) || K < 32) return MI355_EUNSUPPORTED;
    const long big_tiles = (long)cdiv(M, 128) * cdiv(N, 128), tiles = (long)cdiv(M, 16) * cdiv(N, 32);
    const int ncu = resident_slots(1);
    if (big_tiles * 8 > ncu || tiles > 32L * ncu) return MI355_EUNSUPPORTED;       // the engine fills at least an eighth of the chip: its tiles win
    // rejections sit in FRONT of MI355_TRACE (a refused launch must not leave an event pair under this kernel's tag: ADVICE round 5)
    if (precision != MI355_PREC_STRICT && precision != MI355_PREC_FP16 && precision != MI355_PREC_BF16)
        return fail(MI355_EINVAL, "precision must be 0, 1 or 2 (got %d)", precision);
    SmallArgs g{A, B, bias, C, M, N, K, lda, ldb, ldc, cdiv(N, 32)};
    MI355_TRACE(st, "gemm_small_kernel<prec %d> M=%d N=%d K=%d", precision, M, N, K);
    switch (precision) {
        case MI355_PREC_STRICT: gemm_small_kernel<0, 2><<<(int)tiles, 64, 0, st>>>(g); break;
        case MI355_PREC_FP16:   gemm_small_kernel<1, 4><<<(int)tiles, 64, 0, st>>>(g); break;
        case MI355_PREC_BF16:   gemm_small_kernel<2, 4><<<(int)tiles, 64, 0, st>>>(g); break;
        default: return fail(MI355_EINVAL, "precision must be 0, 1 or 2 (got %d)", precision);
    }
    return MI355_OK;
}

}  // namespace mi355
