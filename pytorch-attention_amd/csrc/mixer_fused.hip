// mixer_fused.hip -- the token-mixing half of a MixerLayer in ONE kernel (+ a row-statistics pass), gfx950:
//
//   y = x + ( gelu( LN(x)^T W1^T + b1 ) W2^T + b2 )^T          mlp_mixer.py:47 with Mlp.forward :27-33 and norm1
//
// for N = 196 tokens (14 x 14 patches), T % 32 == 0 hidden token units (the Mixer's default: T = C / 2 = 256), C % 256 == 0 channels.  As three launches
// (transposing LayerNorm -> fc1 + GELU -> transposed-output fc2 + residual: layernorm.hip, gemm16_pa.hip, gemm16.hip) the half moves
// 560 MB and takes 175 us at B = 256, C = 512 for 26 GFLOP: the 16-bit LN(x)^T (59 MB) and the hidden tensor (67 MB) are written and
// read back, and every launch is bound by that traffic.  Here both stay on the chip:
//
//   workgroup = (image, 256 channels), 8 waves; wave = 32 channels (two 16-channel MFMA tiles), ALL tokens
//   phase 1   the wave reads its 32-channel slab of the image (one 128-byte line per token, 8 lanes per line), applies LayerNorm with
//             the row statistics of the pre-pass, and parks the 16-bit result CHANNEL-major in a wave-private LDS region (two tokens
//             per 32-bit word); from there it takes the B operand of the first product -- lane (l15, g): 8 consecutive tokens of
//             channel l15 -- into registers for the whole kernel (14 x 16 bytes)
//   phase 2   hidden units in blocks of 32, W1 / W2 slices double-buffered through LDS (the scheme of mlp_fused_stream_kernel):
//             H^T = W1 . U     (MFMA; a lane then holds 4 consecutive hidden units of one channel) + b1, GELU, re-packed in-lane as
//             the A operand of  O += gelu(H) . W2^T   (rows = channels, columns = tokens)
//   phase 3   a lane holds 4 consecutive CHANNELS of one token per accumulator: + b2[token] + x (re-read: the lines were fetched
//             microseconds ago) and 16-byte stores straight from the accumulators -- the reference's two transposes never exist.
// The stage buffers alias the wave-private regions of phase 1 (one barrier in between).  16-bit operands (fp16 / bf16 per `precision`),
// fp32 accumulation; the GELU result is rounded to 16 bits as the operand of the second product (gelu16_fast4).
#include "common.h"
#include "mma.h"

namespace {

constexpr int MX_N = 196, MX_NK = 224, MX_KS = MX_NK / 32, MX_TMAX = 512, MX_NT = 13, MX_NP = MX_NT * 16;      // T: any multiple of 32 up to TMAX
constexpr int MX_PU = MX_NK + 8;                 // pitch of the parked operand rows and of the W1 slice rows (elements)
constexpr int MX_P2 = 32 + 4;                    // pitch of the W2 slice rows
constexpr int MX_W1S = 32 * MX_PU, MX_W2S = MX_NP * MX_P2, MX_STAGE = MX_W1S + MX_W2S;
constexpr int MX_REGION = 32 * MX_PU;            // elements per wave-private region
constexpr size_t MX_LDS = (size_t)8 * MX_REGION * 2 + (size_t)(MX_TMAX + MX_NP + 2 * MX_NP) * 4;
static_assert((size_t)2 * MX_STAGE * 2 <= (size_t)8 * MX_REGION * 2, "the two weight stages alias the parked operand");
static_assert(MX_LDS <= 160 * 1024, "LDS budget");

struct MixArgs {
    const float* x; float* y; const float* stats;
    const float* ln_w; const float* ln_b;
    const void* w1p;            // (T, 224) 16-bit, columns >= 196 zero
    const void* w2s;            // (T / 32, 208, 32) 16-bit slice-major, rows >= 196 zero
    const float* b1; const float* b2;
    int C, halves, T;           // halves = C / 256 workgroups per image; T hidden token units
    int pair_xcd;               // halves == 2 and B % 8 == 0: the two workgroups of an image get block ids 8 apart (same XCD)
    float eps;                  // LayerNorm eps (STATS kernels)
    unsigned* ovf;              // fp16 range word (common.h rg_report): the 16-bit LN(x) and gelu(H) operands are tracked like the
                                // stand-alone producers they replace (layernorm16_t, the 16-bit GEMM epilogue); null = unguarded
};

// (mean, rstd) per token row: one wave per row, the row in registers (C <= 1024), biased variance, eps inside the sqrt -- the
// statistics of layernorm_kernel.
template <int NV>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int n4 = C >> 2;
    const float inv = 1.0f / (float)C;
    for (long row = wave0; row < rows; row += nwaves) {
        const f4* xr = reinterpret_cast<const f4*>(x + row * C);
        f4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            v[k] = i < n4 ? xr[i] : f4{0.f, 0.f, 0.f, 0.f};
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < n4) { const f4 d = v[k] - mean; q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
        }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = r; }
    }
}

// EARLY: the residual loads of phase 3 are issued in two batches of 14 / 12 ahead of their stores (two exposed round trips instead of
// thirteen: hipcc pairs every token tile's loads with an s_waitcnt vmcnt(0) in front of its stores); the operand registers of phase 2
// are dead by then.  All 26 at once needs 256 registers and spills in the fp16 instantiation.
// STATS: the row statistics of LayerNorm are computed INSIDE the kernel (phase 0) instead of by the row_stats_kernel pre-pass: a
// workgroup holds only half (C = 512) of a token's channels, so each workgroup of an image reads the image's WHOLE rows once more for
// the statistics -- 2 x the kernel's reads, but the second request for a line is served by L2 when the workgroups of an image run on
// one XCD (block ids 8 apart: see the id mapping), and the 23 us pre-pass (103 MB from HBM + a launch) disappears.  Same per-lane sums
// in the same order as row_stats_kernel: bit-identical statistics.  STATS = float4 per lane and row (C / 256; 0 = pre-pass).
template <int PREC, bool EARLY, int STATS>
__global__ __launch_bounds__(512, 2) void mixer_token_kernel(const MixArgs a) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int TT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    el* s_el = reinterpret_cast<el*>(lds);                          // phase 1: 8 regions [32][PU]; phase 2: two stages
    float* s_b1 = reinterpret_cast<float*>(s_el + 8 * MX_REGION);
    float* s_b2 = s_b1 + MX_TMAX;
    float* s_st = s_b2 + MX_NP;                                      // STATS: {mean, rstd} per token (196 x 2 floats)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    // (image, channel half) of this workgroup.  Two halves and a batch that is a multiple of 8: ids 16 k + h * 8 + j belong to image
    // 8 k + j -- the two workgroups of an image are 8 ids apart, i.e. on the same XCD (ids go to the 8 XCDs round-robin) and
    // dispatched together, so whatever one fetches the other finds in that XCD's L2.
    int b, hsel;
    if (a.halves == 2 && a.pair_xcd) {
        const int grp = blockIdx.x >> 4, r = blockIdx.x & 15;
        b = grp * 8 + (r & 7);
        hsel = r >> 3;
    } else {
        b = blockIdx.x / a.halves;
        hsel = blockIdx.x % a.halves;
    }
    const int cw = hsel * 256 + wave * 32;                            // this wave's first channel
    const int C = a.C;
    if (t < a.T) s_b1[t] = a.b1[t];
    if (t < MX_NP) s_b2[t] = t < MX_N ? a.b2[t] : 0.f;
    float rgmax = 0.f;                                              // fp16 range guard: largest finite magnitude this lane converts

    // ---- phase 0 (STATS): mean / rstd of the image's 196 token rows, wave w takes tokens w, w + 8, ...: 13 rows in flight, then 12; the
    //      reductions of a batch are independent chains (their swizzles pipeline) ---------------------------------------------------------
    if constexpr (STATS != 0) {
        constexpr int RB = 13;
        const int n4 = C >> 2;
        const float inv = 1.0f / (float)C;
        const float* xi = a.x + (long)b * MX_N * C;
#pragma unroll 1
        for (int n0 = wave; n0 < MX_N; n0 += 8 * RB) {
            f4 v[RB][STATS];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int n = n0 + 8 * u;
                const f4* xr = reinterpret_cast<const f4*>(xi + (long)(n < MX_N ? n : MX_N - 1) * C);
#pragma unroll
                for (int k = 0; k < STATS; ++k) {
                    const int i = lane + 64 * k;
                    v[u][k] = i < n4 ? xr[i] : f4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float mean[RB], q[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < STATS; ++k) sm += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
                mean[u] = sm;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) mean[u] = wave_sum_sw(mean[u]) * inv;
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                float qq = 0.f;
#pragma unroll
                for (int k = 0; k < STATS; ++k) {
                    if (lane + 64 * k < n4) { const f4 d = v[u][k] - mean[u]; qq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
                }
                q[u] = qq;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) q[u] = wave_sum_sw(q[u]);
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int n = n0 + 8 * u;
                const float r = 1.0f / sqrtf(q[u] * inv + a.eps);
                if (lane == 0 && n < MX_N) { s_st[2 * n] = mean[u]; s_st[2 * n + 1] = r; }
            }
        }
        __syncthreads();
    }

    // ---- phase 1: LayerNorm of the wave's 32-channel slab, parked channel-major (two tokens per word) ----------------------------------
    {
        unsigned int* reg = reinterpret_cast<unsigned int*>(s_el + wave * MX_REGION);     // word (row c, token pair p) at c * (PU / 2) + p
        const int cq = lane & 7, tk = lane >> 3;
        const float* xb0 = a.x + (long)b * MX_N * C + cw + cq * 4;
        const float* st0 = a.stats + (long)b * MX_N * 2;
        const f4 lw = *reinterpret_cast<const f4*>(a.ln_w + cw + cq * 4), lb = *reinterpret_cast<const f4*>(a.ln_b + cw + cq * 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f4 xa[7], xc[7], st[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = (half * 7 + i) * 8 + tk;                 // token pair: tokens 2p, 2p + 1 (N is even: a pair is in or out)
                xa[i] = xc[i] = f4{0.f, 0.f, 0.f, 0.f};
                st[i] = f4{0.f, 0.f, 0.f, 0.f};
                if (2 * p < MX_N) {
                    xa[i] = *reinterpret_cast<const f4*>(xb0 + (long)(2 * p) * C);
                    xc[i] = *reinterpret_cast<const f4*>(xb0 + (long)(2 * p + 1) * C);
                    if constexpr (STATS != 0) st[i] = *reinterpret_cast<const f4*>(s_st + 4 * p);
                    else st[i] = *reinterpret_cast<const f4*>(st0 + 4 * p);    // {mean, rstd} of both tokens
                }
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = (half * 7 + i) * 8 + tk;
                v4 ha = v4{(el)0.f, (el)0.f, (el)0.f, (el)0.f}, hc = ha;
                if (2 * p < MX_N) {
                    const f4 ua = (xa[i] - st[i].x) * st[i].y * lw + lb, uc = (xc[i] - st[i].z) * st[i].w * lw + lb;
                    if constexpr (PREC == 1) rgmax = rg_absmax4(rg_absmax4(rgmax, ua), uc);
                    ha = M_::cvt(ua);
                    hc = M_::cvt(uc);
                }
                auto pack = [](el lo, el hi) {
                    return (unsigned int)__builtin_bit_cast(unsigned short, lo) | ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
                };
                unsigned int* row = reg + (cq * 4) * (MX_PU / 2) + p;
                row[0 * (MX_PU / 2)] = pack(ha.x, hc.x);
                row[1 * (MX_PU / 2)] = pack(ha.y, hc.y);
                row[2 * (MX_PU / 2)] = pack(ha.z, hc.z);
                row[3 * (MX_PU / 2)] = pack(ha.w, hc.w);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // B operand of the first product: lane (l15, g) holds tokens ks*32 + g*8 + [0,8) of channel cw + tt*16 + l15
    v8 xb[TT][MX_KS];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ks = 0; ks < MX_KS; ++ks)
            xb[tt][ks] = *reinterpret_cast<const v8*>(s_el + wave * MX_REGION + (tt * 16 + l15) * MX_PU + ks * 32 + g * 8);

    // ---- phase 2: hidden blocks of 32; W1 rows / W2 columns of the block stream through two LDS stages ---------------------------------
    const el* w1 = static_cast<const el*>(a.w1p);
    const el* w2 = static_cast<const el*>(a.w2s);
    constexpr int W1C = 32 * (MX_NK / 8), W2C = MX_NP * 4;          // 16-byte chunks per slice: 896 and 832
    v8 n1[2], n2[2];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = t + 512 * i;
            if (q < W1C) n1[i] = *reinterpret_cast<const v8*>(w1 + ((long)kb * 32 + q / (MX_NK / 8)) * MX_NK + (q % (MX_NK / 8)) * 8);
            if (q < W2C) n2[i] = *reinterpret_cast<const v8*>(w2 + ((long)kb * MX_NP + (q >> 2)) * 32 + (q & 3) * 8);
        }
    };
    auto commit = [&](int buf) {
        el* d = s_el + buf * MX_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = t + 512 * i;
            if (q < W1C) *reinterpret_cast<v8*>(d + (q / (MX_NK / 8)) * MX_PU + (q % (MX_NK / 8)) * 8) = n1[i];
            if (q < W2C) {
                el* p = d + MX_W1S + (q >> 2) * MX_P2 + (q & 3) * 8;     // rows are 72 bytes apart: two 8-byte stores
                *reinterpret_cast<v4*>(p) = v4{n2[i][0], n2[i][1], n2[i][2], n2[i][3]};
                *reinterpret_cast<v4*>(p + 4) = v4{n2[i][4], n2[i][5], n2[i][6], n2[i][7]};
            }
        }
    };
    f4 o[TT][MX_NT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int nt = 0; nt < MX_NT; ++nt) o[tt][nt] = f4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    __syncthreads();                                               // every wave has taken its operand out of the regions the stages alias
    commit(0);
    __syncthreads();
    const int nkb = a.T >> 5;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) fetch(kb + 1);                        // in flight during the MFMAs below
        const el* sw1 = s_el + buf * MX_STAGE;
        const el* sw2 = sw1 + MX_W1S;
        f4 s[TT][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f4 bias = *reinterpret_cast<const f4*>(s_b1 + kb * 32 + h2 * 16 + g * 4);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) s[tt][h2] = bias;
#pragma unroll
            for (int ks = 0; ks < MX_KS; ++ks) {
                const v8 wf = *reinterpret_cast<const v8*>(sw1 + (h2 * 16 + l15) * MX_PU + ks * 32 + g * 8);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) s[tt][h2] = M_::mma(wf, xb[tt][ks], s[tt][h2]);
            }
        }
        v8 pf[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const f4 p0 = gelu16_fast4(s[tt][0]);
            const f4 p1 = gelu16_fast4(s[tt][1]);
            if constexpr (PREC == 1) rgmax = rg_absmax4(rg_absmax4(rgmax, p0), p1);
            const v4 h0 = M_::cvt(p0), h1 = M_::cvt(p1);
            pf[tt] = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        }
#pragma unroll
        for (int nt = 0; nt < MX_NT; ++nt) {
            const el* wr = sw2 + (nt * 16 + l15) * MX_P2 + g * 4;
            const v4 a0 = *reinterpret_cast<const v4*>(wr), a1 = *reinterpret_cast<const v4*>(wr + 16);
            const v8 vf = v8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) o[tt][nt] = M_::mma(pf[tt], vf, o[tt][nt]);
        }
        if (kb + 1 < nkb) {
            commit(buf ^ 1);                                       // stage buf^1 was last read at block kb-1: everybody passed the barrier below
            __syncthreads();
        }
    }
    // ---- phase 3: lane (l15, g) holds channels cw + tt*16 + g*4 + [0,4) of token nt*16 + l15: + b2 + x, 16-byte stores -----------------
    const long base = (long)b * MX_N * C + cw + g * 4;
    if constexpr (EARLY) {
        constexpr int CH = 7;                                          // token tiles per batch: 14 loads in flight, then their stores
#pragma unroll
        for (int c0 = 0; c0 < MX_NT; c0 += CH) {
            f4 xr[TT][CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int n = (c0 + i) * 16 + l15;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
                    xr[tt][i] = (c0 + i < MX_NT && n < MX_N) ? *reinterpret_cast<const f4*>(a.x + base + (long)n * C + tt * 16) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int nt = c0 + i, n = nt * 16 + l15;
                if (nt >= MX_NT || n >= MX_N) continue;
                const float bn = s_b2[n];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<f4*>(a.y + base + (long)n * C + tt * 16) = (o[tt][nt] + bn) + xr[tt][i];
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < MX_NT; ++nt) {
            const int n = nt * 16 + l15;
            if (n >= MX_N) continue;
            const float bn = s_b2[n];
            f4 xr[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) xr[tt] = *reinterpret_cast<const f4*>(a.x + base + (long)n * C + tt * 16);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<f4*>(a.y + base + (long)n * C + tt * 16) = (o[tt][nt] + bn) + xr[tt];
        }
    }
    if constexpr (PREC == 1) rg_report(rgmax, a.ovf, 4u);
}

}  // namespace

extern "C" {

size_t mi355_mixer_token_workspace_bytes(int B, int N, int C) {
    (void)C;
    if (B <= 0 || N <= 0) return 16;
    return (size_t)B * N * 2 * sizeof(float) + 16;
}

int mi355_mixer_token_fwd(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const void* w1p16, const float* b1,
                          const void* w2s16, const float* b2, float* y, int B, int N, int C, int T, int precision, void* ws, size_t ws_bytes,
                          mi355_stream_t stream) {
    MI355_CHECK_ARG(x && ln_w && ln_b && w1p16 && b1 && w2s16 && b2 && y && ws && B > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (N != MX_N || T <= 0 || (T % 32) != 0 || T > MX_TMAX || C <= 0 || (C % 256) != 0 || C > 1024)
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_mixer_token_fwd: built for N = 196 tokens, T %% 32 == 0 (<= 512), C %% 256 == 0 (<= 1024) (N=%d T=%d C=%d)", N, T, C);
    if (!aligned16(x) || !aligned16(y) || !aligned16(ws) || !aligned16(w1p16) || !aligned16(w2s16) || !aligned16(ln_w) || !aligned16(ln_b))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_mixer_token_fwd: 16-byte aligned buffers required");
    MI355_CHECK_ARG(ws_bytes >= mi355_mixer_token_workspace_bytes(B, N, C));
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* stats = static_cast<float*>(ws);
    const long rows = (long)B * N;
    const bool instats = mi355::opt_mixer_stats() != 0 && C == 512;    // statistics inside the token kernel (phase 0; built for C = 512) or by the pre-pass
    if (!instats) {
        const int sgrid = (int)(cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192);
        MI355_TRACE(st, "row_stats_kernel rows=%ld cols=%d", rows, C);
        if (C <= 256) row_stats_kernel<1><<<sgrid, 256, 0, st>>>(x, stats, rows, C, ln_eps);
        else if (C <= 512) row_stats_kernel<2><<<sgrid, 256, 0, st>>>(x, stats, rows, C, ln_eps);
        else row_stats_kernel<4><<<sgrid, 256, 0, st>>>(x, stats, rows, C, ln_eps);
        MI355_LAUNCH_CHECK();
    }
    MixArgs a{};
    a.x = x; a.y = y; a.stats = stats; a.ln_w = ln_w; a.ln_b = ln_b; a.w1p = w1p16; a.w2s = w2s16; a.b1 = b1; a.b2 = b2;
    a.C = C; a.halves = C / 256; a.T = T; a.eps = ln_eps;
    a.pair_xcd = (a.halves == 2 && (B % 8) == 0) ? 1 : 0;
    const long grid = (long)B * a.halves;
    if (grid >= (1L << 31)) return mi355::fail(MI355_EUNSUPPORTED, "mi355_mixer_token_fwd: batch too large");
    a.ovf = precision == MI355_PREC_FP16 ? mi355::range_word(st) : nullptr;
    const bool early = mi355::opt_mixer_early() != 0;
    MI355_TRACE(st, "mixer_token_kernel%s%s B=%d C=%d", early ? "<early>" : "", instats ? "<stats>" : "", B, C);
#define MIXER_LAUNCH(P_, E_, S_)                                                                                                      \
    do {                                                                                                                             \
        if (int rc = mi355::func_dynamic_lds(reinterpret_cast<const void*>(mixer_token_kernel<P_, E_, S_>), (int)MX_LDS)) return rc;  \
        mixer_token_kernel<P_, E_, S_><<<(int)grid, 512, MX_LDS, st>>>(a);                                                           \
    } while (0)
#define MIXER_BY(P_)                                                                            \
    do {                                                                                        \
        if (early) { if (instats) MIXER_LAUNCH(P_, true, 2); else MIXER_LAUNCH(P_, true, 0); }   \
        else       { if (instats) MIXER_LAUNCH(P_, false, 2); else MIXER_LAUNCH(P_, false, 0); } \
    } while (0)
    if (precision == MI355_PREC_FP16) MIXER_BY(1);
    else MIXER_BY(2);
#undef MIXER_BY
#undef MIXER_LAUNCH
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
