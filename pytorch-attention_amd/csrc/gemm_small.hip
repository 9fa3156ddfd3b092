// gemm_small.hip -- nn.Linear on SMALL outputs (the classifier head of a ViT: ViT.py:191, M = 256 images, N = 1000 classes, K = 768).
//
// The fp32-in engine (gemm.hip) cuts C into 128 x 128 tiles: 16 workgroups for the ViT head -- 240 of the 256 CUs idle, 47-97 us for a
// 0.39 GFLOP product ("What's missing" 6).  Here a tile is 32 x 32 and belongs to ONE WAVE (a 64-thread workgroup):
// 256 workgroups for the head, one per CU.  No LDS at all: a lane loads the 8 consecutive fp32 k-values of its fragment row straight from
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
    const int m0 = tm * 32, n0 = tn * 32;
    // fragment rows of this lane (clamped: rows past the edge load a valid row and are never stored)
    const float* ap[2];
    const float* bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + frow, n = n0 + i * 16 + frow;
        ap[i] = g.A + (long)(m < g.M ? m : g.M - 1) * g.lda + fk;
        bp[i] = g.B + (long)(n < g.N ? n : g.N - 1) * g.ldb + fk;
    }
    f4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    auto pack = [](f4 lo4, f4 hi4, v8* out) {                      // 8 fp32 -> operand fragment (hi, and lo in strict mode)
        const v4 a = M_::cvt(lo4), b = M_::cvt(hi4);
        out[0] = v8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if constexpr (NS == 2) {
            const v4 c = M_::cvt_lo(lo4, a), d = M_::cvt_lo(hi4, b);
            out[1] = v8{c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        }
    };
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < g.K; k0 += 32 * CH) {
        f4 ra[CH][2][2], rb[CH][2][2];                              // [k step][fragment][half of the 8 k-values]
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int k = k0 + s * 32 + fk;                          // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[s][i][0] = k < g.K ? *reinterpret_cast<const f4*>(ap[i] + k0 + s * 32) : zero;
                ra[s][i][1] = k + 4 < g.K ? *reinterpret_cast<const f4*>(ap[i] + k0 + s * 32 + 4) : zero;
                rb[s][i][0] = k < g.K ? *reinterpret_cast<const f4*>(bp[i] + k0 + s * 32) : zero;
                rb[s][i][1] = k + 4 < g.K ? *reinterpret_cast<const f4*>(bp[i] + k0 + s * 32 + 4) : zero;
            }
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (k0 + s * 32 >= g.K) break;                          // whole step past K (uniform): nothing to add
            v8 fa[2][NS], fb[2][NS];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pack(ra[s][i][0], ra[s][i][1], fa[i]);
                pack(rb[s][i][0], rb[s][i][1], fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_step<PREC>(fa[i], fb[j], acc[i][j]);
        }
    }
    // C / D layout: lane holds column (lane & 15), rows (lane >> 4) * 4 + [0, 4) of each 16 x 16 tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + frow;
        if (n >= g.N) continue;
        const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                if (m < g.M) g.C[(long)m * g.ldc + n] = g.bias ? acc[i][j][r] + bn : acc[i][j][r];
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
    const long big_tiles = (long)cdiv(M, 128) * cdiv(N, 128), tiles = (long)cdiv(M, 32) * cdiv(N, 32);
    const int ncu = resident_slots(1);
    if (big_tiles * 4 > ncu || tiles > 16L * ncu) return MI355_EUNSUPPORTED;       // the engine fills at least a quarter of the chip: its tiles win
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
