// mlp_wide.hip -- y = x + gamma * (W2 gelu(W1' LN(x) + b1') + b2) in ONE kernel for token streams of width C = 256 / 384 (hidden 4C):
// the MLP half of a CSWin stage-3 block (cswin.py:194-196) and of an XCiT-S XCABlock (xcit.py:294), gfx950.  Round 6.
//
// Unfused, this half is three launches -- LayerNorm, fc1 + GELU, fc2 + residual -- that move the 16-bit hidden tensor (M x 4C: 154 MB at
// C = 384, B = 256) out to HBM and back and pay two short-K GEMM epilogues: 224 us of the 469-us XCABlock, 134 of the 265-us CSWin stage-3
// block (profiles/r06_*_kernel_seq.txt).  mlp_fused.hip stops at C = 128 because its waves split the TOKENS and every wave reads every
// weight fragment from LDS.  Here the waves split the WEIGHTS, and the two products run on DIFFERENT waves:
//
//   workgroup  = 8 waves, persistent, one per CU; a step = R <= 112 token rows (7 row tiles of 16; the launcher picks R so that the steps
//                fill whole rounds: B = 256 x 196 tokens on 256 CUs = 512 steps of 98 rows);
//   per step   : the rows are LayerNorm'ed in registers (a row per wave pass) and parked in LDS once, 16 bit (s_xn, 88 KB at C = 384);
//   waves 0-3  PRODUCERS, per slice of 128 hidden units:  H^T = W1' . Xn^T for the wave's 32 hidden units and ALL rows.  Its W1' rows are
//              needed by no other wave, so the fragments go global -> VGPR (16-byte loads along K, two k-steps ahead of their MFMAs) and never
//              touch LDS; an Xn fragment read from LDS feeds two MFMAs.  + b1', GELU, 16 bit -> s_h[slice & 1][row][128];
//   waves 4-7  CONSUMERS, one slice behind:  Y^T += W2 . gelu(H)^T for the wave's C/4 output channels and all rows, W2 fragments
//              global -> VGPR, a gelu(H) fragment read from LDS feeds C/64 MFMAs;
//   ONE barrier per slice hands buffer (slice & 1) over.  Every SIMD hosts one producer and one consumer wave: the GELU of slice i (VALU)
//   runs under the second product of slice i - 1 (matrix pipe) -- the version with all eight waves in both roles (two barriers per 256
//   hidden units, GELU between them with the matrix pipe idle) measured 242 us at C = 384 against 224 us for the three launches;
//   epilogue   (consumers) (+ b2) * gamma + x (re-read: L2 / Infinity-Cache resident) -> 16-byte stores of C-byte row segments.
// The weights cross L2 -> CU once per step (2.36 MB at C = 384): 512 steps x 2.36 MB = 1.2 GB at the ~10 TB/s the chip's CUs pull together
// is the kernel's floor.  LayerNorm's affine part is folded into W1 / b1 by the caller (W1' = W1 diag(ln_w), b1' = b1 + W1 ln_b); weights
// 16-bit (fp16 / bf16 per `precision`), accumulation fp32, the hidden activation is rounded to 16 bit exactly where the unfused path
// rounds it.
#include "common.h"
#include "mma.h"

namespace {

struct WideArgs {
    const float* x; float* y;
    const void* w1; const void* w2;            // 16-bit: (4C, C) and (C, 4C), row-major
    const float* b1; const float* b2; const float* gamma;
    long M; long nsteps;
    int rows;                                  // rows per step (<= 112)
    float eps;
    int do_ln;
    unsigned* ovf;                             // fp16 range word (code 4) or null
};

template <int PREC, int C>
__global__ __launch_bounds__(512, 1) void mlp_wide_kernel(const WideArgs a) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int HD = 4 * C, NW = 8, RT = 7, RS = RT * 16;
    constexpr int KS = C / 32;                 // k-steps of the first product
    constexpr int NT = C / 64;                 // 16-column output tiles per consumer wave
    constexpr int NCW = NT * 16;
    constexpr int HS = 128, NSL = HD / HS, KS2 = HS / 32;
    constexpr int XP = C + 8, HP = HS + 8;     // LDS row pitches (elements)
    constexpr int FPL = C / 64;                // floats of a row per lane in the LayerNorm pass (4 or 6)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    el* s_xn = reinterpret_cast<el*>(lds);
    el* s_h = s_xn + RS * XP;                  // two buffers of RS x HP
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const bool producer = wave < 4;            // wave-uniform
    const int rw = wave & 3;                   // index inside the role
    const el* __restrict__ w1 = static_cast<const el*>(a.w1);
    const el* __restrict__ w2 = static_cast<const el*>(a.w2);
    const float invC = 1.0f / (float)C;
    float rgmax = 0.f;

    for (long step = blockIdx.x; step < a.nsteps; step += gridDim.x) {
        const long row0 = step * a.rows;
        const long left = a.M - row0;
        const int rows = (int)(left < a.rows ? left : a.rows);
        // ---- LayerNorm (two-pass, biased variance, eps inside the sqrt: the statistics of layernorm_kernel) -> s_xn, 16 bit -------------
        for (int r = wave; r < RS; r += NW) {
            float v[FPL];
            if (r < rows) {
                const float* xr = a.x + (row0 + r) * C + lane * FPL;
#pragma unroll
                for (int j = 0; j < FPL; j += 2) {
                    const float2 p = *reinterpret_cast<const float2*>(xr + j);
                    v[j] = p.x; v[j + 1] = p.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < FPL; ++j) v[j] = 0.f;
            }
            float mean = 0.f, rstd = 1.f;
            if (a.do_ln) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < FPL; ++j) s += v[j];
                mean = wave_sum(s) * invC;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < FPL; ++j) { const float d = v[j] - mean; q += d * d; }
                rstd = 1.0f / sqrtf(wave_sum(q) * invC + a.eps);
            }
            el* dst = s_xn + r * XP + lane * FPL;
#pragma unroll
            for (int j = 0; j < FPL; j += 2) {
                typedef el e2 __attribute__((ext_vector_type(2)));
                const float n0 = r < rows ? (v[j] - mean) * rstd : 0.f, n1 = r < rows ? (v[j + 1] - mean) * rstd : 0.f;
                *reinterpret_cast<e2*>(dst + j) = e2{M_::cvt1(n0), M_::cvt1(n1)};
            }
        }
        __syncthreads();

        if (producer) {
            // =========== H^T = W1' Xn^T, + b1', GELU -> s_h[sl & 1]: 32 hidden units per wave and slice, all rows ==========================
#pragma unroll 1
            for (int it = 0; it <= NSL; ++it) {
                if (it < NSL) {
                    const int h0 = it * HS + rw * 32;
                    f4 s[RT][2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const f4 bias = *reinterpret_cast<const f4*>(a.b1 + h0 + h2 * 16 + g * 4);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) s[rt][h2] = bias;
                    }
                    const el* w1p = w1 + (long)(h0 + l15) * C + g * 8;       // A fragment of hidden tile h2 at k-step ks: + h2 * 16 * C + ks * 32
                    v8 wf[3][2];
#pragma unroll
                    for (int pre = 0; pre < 2; ++pre)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) wf[pre][h2] = *reinterpret_cast<const v8*>(w1p + (long)h2 * 16 * C + pre * 32);
                    // Xn fragments in a flat pipeline over (ks, rt), three reads ahead of their MFMAs
                    v8 xf[4];
#pragma unroll
                    for (int pre = 0; pre < 3; ++pre) xf[pre] = *reinterpret_cast<const v8*>(s_xn + (pre * 16 + l15) * XP + g * 8);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) {
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) wf[(ks + 2) % 3][h2] = *reinterpret_cast<const v8*>(w1p + (long)h2 * 16 * C + (ks + 2) * 32);
                        }
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int idx = ks * RT + rt, nx = idx + 3;
                            if (nx < KS * RT)
                                xf[nx & 3] = *reinterpret_cast<const v8*>(s_xn + ((nx % RT) * 16 + l15) * XP + (nx / RT) * 32 + g * 8);
                            s[rt][0] = M_::mma(wf[ks % 3][0], xf[idx & 3], s[rt][0]);
                            s[rt][1] = M_::mma(wf[ks % 3][1], xf[idx & 3], s[rt][1]);
                        }
                    }
                    // GELU, 16 bit, parked row-major: lane (l15, g) holds hidden units h2 * 16 + g * 4 + [0,4) of row rt * 16 + l15
                    el* hb = s_h + (it & 1) * (RS * HP);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const f4 p = gelu16_fast4(s[rt][h2]);
                            if constexpr (PREC == 1) rgmax = rg_max3abs4(rgmax, p);
                            *reinterpret_cast<v4*>(hb + (rt * 16 + l15) * HP + rw * 32 + h2 * 16 + g * 4) = M_::cvt(p);
                        }
                }
                __syncthreads();                                     // slice `it` is complete in s_h[it & 1]; the consumers are done with slice it - 2
            }
        } else {
            // =========== Y^T += W2 gelu(H)^T, one slice behind the producers: C/4 output channels per wave, all rows ==========================
            f4 o[RT][NT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) o[rt][ct] = f4{0.f, 0.f, 0.f, 0.f};
            const el* w2p = w2 + (long)(rw * NCW + l15) * HD + g * 8;        // A fragment of column tile ct at hidden offset hoff: + ct * 16 * HD + hoff
            v8 vcur[NT], vnxt[NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) vcur[ct] = *reinterpret_cast<const v8*>(w2p + (long)ct * 16 * HD);
            __syncthreads();                                         // it = 0: nothing to consume yet
#pragma unroll 1
            for (int sl = 0; sl < NSL; ++sl) {
                const el* hb = s_h + (sl & 1) * (RS * HP) + l15 * HP + g * 8;
                // gelu(H) fragments: a pipeline over (k2, rt), two reads ahead of their MFMAs, rotated through three registers
                v8 p0 = *reinterpret_cast<const v8*>(hb), p1 = *reinterpret_cast<const v8*>(hb + 16 * HP), p2;
#pragma unroll 1
                for (int k2 = 0; k2 < KS2; ++k2) {
                    // the W2 fragments of the next k-step (of the next slice behind the last one) are requested a k-step ahead
                    const int hoff = sl * HS + (k2 + 1) * 32;        // == (sl + 1) * HS when k2 + 1 == KS2
                    if (hoff < HD) {
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) vnxt[ct] = *reinterpret_cast<const v8*>(w2p + (long)ct * 16 * HD + hoff);
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int nrt = (rt + 2) % RT, dk = (rt + 2) / RT;          // compile-time
                        if (dk == 0 || k2 + 1 < KS2) p2 = *reinterpret_cast<const v8*>(hb + nrt * 16 * HP + (k2 + dk) * 32);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) o[rt][ct] = M_::mma(vcur[ct], p0, o[rt][ct]);
                        p0 = p1; p1 = p2;
                        __builtin_amdgcn_sched_barrier(0);           // keep the reads two ahead, not seven: the accumulators leave no room
                    }
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) vcur[ct] = vnxt[ct];
                }
                __syncthreads();
            }
            // ---- epilogue: (+ b2) * gamma + x -> y; lane (l15, g) holds channels rw * NCW + ct * 16 + g * 4 + [0,4) of row rt * 16 + l15.
            //      The residual rows of two column tiles are requested together (one exposed L2 round trip per pair) ---------------------------
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                f4 xr[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = rt * 16 + l15;
                    xr[rt] = f4{0.f, 0.f, 0.f, 0.f};
                    if (r < rows) xr[rt] = *reinterpret_cast<const f4*>(a.x + (row0 + r) * C + rw * NCW + ct * 16 + g * 4);
                }
                const int c0 = rw * NCW + ct * 16 + g * 4;
                const f4 b2 = a.b2 ? *reinterpret_cast<const f4*>(a.b2 + c0) : f4{0.f, 0.f, 0.f, 0.f};
                const f4 gm = a.gamma ? *reinterpret_cast<const f4*>(a.gamma + c0) : f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r = rt * 16 + l15;
                    if (r < rows) {
                        f4 v = o[rt][ct] + b2;
                        if (a.gamma) v = v * gm;
                        *reinterpret_cast<f4*>(a.y + (row0 + r) * C + c0) = v + xr[rt];
                    }
                }
            }
        }
        // the next step's LayerNorm overwrites s_xn: the producers' last read of it lies in front of the last slice barrier, which every wave
        // has passed; s_h[0] is written again only behind the barrier that follows that LayerNorm
    }
    if constexpr (PREC == 1) rg_report_f(rgmax, a.ovf, 4u);
}

template <int C>
constexpr size_t wide_smem() { return (size_t)112 * (C + 8) * 2 + (size_t)2 * 112 * (128 + 8) * 2; }

}  // namespace

namespace mi355 {

bool mlp_wide_applicable(int C, int hidden) { return (C == 256 || C == 384) && hidden == 4 * C; }

// layernorm: bit 0 = normalise, bit 1 = the caller proved |gelu(H)| < 65504 from the folded weights (see mi355_mlp_fused_fwd)
int mlp_wide(const float* x, const void* w1_16, const float* b1, const void* w2_16, const float* b2, const float* gamma, float* y, long M, int C,
             int layernorm, float eps, int precision, hipStream_t st) {
    WideArgs a{};
    a.x = x; a.y = y; a.w1 = w1_16; a.w2 = w2_16; a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.M = M; a.eps = eps; a.do_ln = layernorm & 1;
    const int ncu = resident_slots(1);
    // whole rounds: k = the number of rounds that 112-row steps need, then the rows are spread evenly over k * ncu steps
    const long k = (M + 112L * ncu - 1) / (112L * ncu);
    long rows = (M + k * ncu - 1) / (k * ncu);
    if (rows < 16) rows = M < 16 ? M : 16;
    if (rows > 112) rows = 112;
    a.rows = (int)rows;
    a.nsteps = (M + rows - 1) / rows;
    a.ovf = (precision == MI355_PREC_FP16 && !(layernorm & 2)) ? range_word(st) : nullptr;
    const int grid = (int)(a.nsteps < ncu ? a.nsteps : ncu);
    MI355_TRACE(st, "mlp_wide_kernel<C=%d> M=%ld rows/step=%d", C, M, a.rows);
#define GO(P_, C_)                                                                                                              \
    do {                                                                                                                        \
        constexpr size_t sm = wide_smem<C_>();                                                                                  \
        static_assert(sm <= 160 * 1024, "LDS budget");                                                                          \
        if (int rc = func_dynamic_lds(reinterpret_cast<const void*>(mlp_wide_kernel<P_, C_>), (int)sm)) return rc;              \
        mlp_wide_kernel<P_, C_><<<grid, 512, sm, st>>>(a);                                                                      \
    } while (0)
    if (precision == MI355_PREC_FP16) { if (C == 256) GO(1, 256); else GO(1, 384); }
    else                              { if (C == 256) GO(2, 256); else GO(2, 384); }
#undef GO
    return MI355_OK;
}

}  // namespace mi355
