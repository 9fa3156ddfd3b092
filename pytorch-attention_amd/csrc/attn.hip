// attn.hip -- fused QK^T-softmax-PV core for short sequences / stripe windows on gfx950 (MFMA 16x16x32).
//
// Serves  * ViT Attention core        (ViT.py:82-86):      one "window" = the whole sequence (N = 197), d = 64
//         * CSWin LePEAttention       (cswin.py:101-127):  stripe windows of 49..98 tokens, d = 32, + LePE
//
// One workgroup (4 waves) per (image, window, head).  K and V of the head are converted to the MFMA operand format
// and parked in LDS (K as [key][d], V transposed as [d][key], both K-contiguous for their MFMA); every wave then owns
// whole 16-query tiles:
//     S^T = K . Q^T      (MFMA, A = K tile from LDS, B = Q fragment straight from HBM)      scores never leave registers
//     softmax over keys  (keys of one query sit in 4 registers x KT tiles x 4 lane groups: in-lane max/sum + 2 xor-shuffles)
//     O   = P . V        (MFMA, A = P re-packed IN-LANE -- the S^T accumulator layout is already the A-operand layout
//                         once 32-key blocks are enumerated as (tile 2kb | tile 2kb+1) x (4 lane groups) x 4 --, B = V^T from LDS)
//     O  /= rowsum, (+ LePE: depth-wise 3x3 over the window image of V, zero padded at the window border), staged through a
//     per-wave LDS slab and written as whole 128/256-byte rows at their window-scattered token positions.
// The window partition / head split / merge of the reference (img2windows, windows2img, reshape+permute chains) is pure
// index math here: token slot t of window w is token l = ((w / nWx) * Hsp + t / Wsp) * reso + (w % nWx) * Wsp + t % Wsp.
//
// HBM traffic per call = read qkv once + write out once (the isolated core is HBM-bound: SURVEY.md 8d).
#include "common.h"
#include "mma.h"
#include "bufops.h"
#include <type_traits>

namespace {

struct AttnArgs {
    const void* qkv;       // (B, L, 3, Ctot)   fp32, or the 16-bit operand type when IO16
    void* out;             // (B, L, Ctot)      same element type as qkv
    const float* lepe_w;   // (Cb, 3, 3) or null
    const float* lepe_b;   // (Cb)
    int L, Ctot, c0, heads;
    int koff, voff, oc0;   // element offsets from a head's q slice to its k / v slices inside a token row; first output channel of the branch
    int reso, Hsp, Wsp, nWx, nwin;   // window geometry on the token grid
    int T;                 // tokens per window
    unsigned wsp_magic;    // ceil(2^32 / Wsp): slot / Wsp == umulhi(slot, magic) for slot, Wsp < 2^16 (no integer divide in the kernel)
    float scale;
    int pre_scale;         // 1: q*scale before QK^T (CSWin), 0: (QK^T)*scale (ViT)
};

// Two argument sets per launch: the two stripe branches of a CSWinBlock (cswin.py:155-165: vertical and horizontal stripes on the two
// channel halves) have the same tile shape and run as ONE grid -- blocks [0, split) take a0, the rest a1 -- so the chip is filled
// once instead of twice (the per-block critical path is ~18 us; two half-size launches each paid their own ramp and tail).
struct AttnPair {
    AttnArgs a0, a1;
    int split;
};

// OCC: workgroups per CU the register allocation is asked to leave room for.  The window kernels are latency-bound (a workgroup does
// one memory round trip, a few dozen MFMAs, one store), so resident workgroups per CU are the throughput lever.
// TFULL: number of key tiles known at compile time to lie entirely below T (the hot shapes are dispatched with it: 197 -> 12,
// 98 -> 6, 56 / 49 -> 3); their scores skip the key-validity mask (two of ~17 VALU instructions per score in a VALU-bound kernel).
// -1 = unknown, every tile is masked.  (Eight key tiles with LePE need 88-96 registers: five workgroups per CU is what the allocator
// reaches, so that is what is asked for; the build treats an unmet occupancy hint as an error.)
#ifndef MI355_ATTN_FMA_SM
#define MI355_ATTN_FMA_SM 1
#endif
template <int PREC, int D, int KT, bool LEPE, bool IO16, int NW, int OCC = 1, int TFULL = -1>
__global__ __launch_bounds__(NW * 64, OCC) void win_attn_kernel(const AttnPair pr) {
    constexpr bool FMA_SM = MI355_ATTN_FMA_SM != 0;
    // XCD-aware block order: hardware hands consecutive block ids to the 8 XCDs round-robin, but consecutive LOGICAL ids are the heads
    // of one window, whose q / k / v slices are adjacent 64-byte (d = 32) pieces of the same token rows -- neighbours that should
    // meet in one XCD's L2.  XCD k therefore works on the contiguous logical range [k * n/8, (k+1) * n/8) (bijective for any n).
    const int lid = xcd_contiguous_block();
    const bool second = lid >= pr.split;
    const AttnArgs a = second ? pr.a1 : pr.a0;
    constexpr int NTHR = NW * 64;             // NW waves share one head's K / V (8 for the 197-token ViT case: 4 waves per SIMD)
    static_assert(!IO16 || PREC != 0, "16-bit I/O exists for the fp16 / bf16 operand modes only");
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int NS = M_::NSPLIT;
    constexpr int TK = KT * 16;               // padded key count
    constexpr int KP = D + 8;                 // K row pitch (elements)
    constexpr int VP = TK + 4;                // V^T row pitch (elements, multiple of 4 -> 8-byte aligned reads)
    constexpr int OP = IO16 ? D + 8 : D + 4;  // O slab pitch (elements: 16-bit when the output is 16-bit, else floats)
    constexpr int K_EL = TK * KP, V_EL = D * VP;
    using slab_t = typename std::conditional<IO16, unsigned short, float>::type;
    __shared__ __attribute__((aligned(16))) unsigned short s_k[NS * K_EL];
    __shared__ __attribute__((aligned(16))) unsigned short s_v[NS * V_EL];
    __shared__ __attribute__((aligned(16))) slab_t s_o[NW * 16 * OP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int bid = lid - (second ? pr.split : 0);
    const int head = bid % a.heads; bid /= a.heads;
    const int win = bid % a.nwin;
    const int b = bid / a.nwin;
    const int T = a.T;
    const int wy0 = (win / a.nWx) * a.Hsp, wx0 = (win % a.nWx) * a.Wsp;
    const int ch0 = a.c0 + head * D;                                  // first element of this head's q slice inside a token row
    const int och0 = a.oc0 + head * D;                                // first channel of this head in the output row
    const long row3 = 3L * a.Ctot;
    using gel = typename std::conditional<IO16, el, float>::type;        // element type in HBM
    const gel* base = static_cast<const gel*>(a.qkv) + (long)b * a.L * row3 + ch0;
    auto srow = [&](int s) { return a.Wsp == 1 ? s : (int)__umulhi((unsigned)s, a.wsp_magic); };   // s / Wsp (magic overflows at 1)
    auto tok = [&](int s) { const int r = srow(s); return (wy0 + r) * a.reso + wx0 + (s - r * a.Wsp); };   // window slot -> token

    const int l15 = lane & 15, g = lane >> 4;
    const int nqt = (T + 15) >> 4;
    // ---- everything this wave will need from HBM is requested up front (one memory latency per workgroup): its Q fragments
    //      (16-bit I/O path), its LePE taps, then the K / V staging loads below --------------------------------------------------
    constexpr int NQ = (KT + NW - 1) / NW;                         // query tiles per wave
    // 16-bit I/O: q / k / v of this image go through one raw buffer descriptor; slots past T get an out-of-range offset and come back
    // as zeros from the hardware range check (no zero-fill moves, no branches around the loads)
    const rsrc_t img_rs = make_rsrc(IO16 ? static_cast<const void*>(base) : nullptr, IO16 ? (bufops_u32)((long)a.L * row3 * sizeof(gel)) : 0u);
    auto ld16 = [&](bool live, int token, int el_off) -> v8 {      // 8 elements of a token row, el_off relative to this head's q slice
        const bufops_u32 off = live ? (bufops_u32)((token * (int)row3 + el_off) * 2) : OOB;
        return __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(img_rs, off, 0, 0));
    };
    v8 qraw[IO16 ? NQ : 1][D / 32];
    if constexpr (IO16) {
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) {
            const int qs = (wave + iq * NW) * 16 + l15;
            const int tq = tok(qs < T ? qs : 0);
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) qraw[iq][ks] = ld16(qs < T, tq, ks * 32 + g * 8);
        }
    }
    // LePE taps + bias of this head's D channels: parked in LDS ([channel][10]) instead of 20 registers per lane
    __shared__ float s_lw[LEPE ? D * 10 : 1];
    if constexpr (LEPE) {
        for (int q = t; q < D * 10; q += NTHR) {
            const int c = q / 10, i = q - c * 10;
            s_lw[q] = i < 9 ? a.lepe_w[(long)(head * D + c) * 9 + i] : a.lepe_b[head * D + c];
        }
    }

    // ---- phase A: K -> LDS [key][d],  V -> LDS transposed [d][key] -------------------------------------------------
    constexpr int D4 = D / 4;
    if constexpr (IO16) {
        // all global loads of the head are issued back to back into registers (one memory latency for K and V together),
        // then written to LDS: K as is, V through a 4(key) x 8(d) register transpose
        constexpr int D8 = D / 8;
        constexpr int NKI = (TK * D8 + NTHR - 1) / NTHR, NVI = ((TK / 4) * D8 + NTHR - 1) / NTHR;
        v8 kreg[NKI];
        v8 vreg[NVI][4];
#pragma unroll
        for (int it = 0; it < NKI; ++it) {
            const int idx = t + it * NTHR, key = idx / D8, d8 = idx % D8;
            const bool live = idx < TK * D8 && key < T;
            kreg[it] = ld16(live, tok(live ? key : 0), a.koff + d8 * 8);
        }
#pragma unroll
        for (int it = 0; it < NVI; ++it) {
            const int idx = t + it * NTHR, kg = idx / D8, d8 = idx % D8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kg * 4 + j;
                const bool live = idx < (TK / 4) * D8 && key < T;
                vreg[it][j] = ld16(live, tok(live ? key : 0), a.voff + d8 * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < NKI; ++it) {
            const int idx = t + it * NTHR, key = idx / D8, d8 = idx % D8;
            if (idx < TK * D8) *reinterpret_cast<v8*>(s_k + key * KP + d8 * 8) = kreg[it];
        }
#pragma unroll
        for (int it = 0; it < NVI; ++it) {
            const int idx = t + it * NTHR, kg = idx / D8, d8 = idx % D8;
            if (idx < (TK / 4) * D8) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *reinterpret_cast<v4*>(s_v + (d8 * 8 + q) * VP + kg * 4) =
                        v4{vreg[it][0][q], vreg[it][1][q], vreg[it][2][q], vreg[it][3][q]};
            }
        }
    } else {
    for (int idx = t; idx < TK * D4; idx += NTHR) {
        const int key = idx / D4, d4 = idx % D4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (key < T) v = *reinterpret_cast<const f4*>(base + (long)tok(key) * row3 + a.koff + d4 * 4);
        const v4 h = M_::cvt(v);
        *reinterpret_cast<v4*>(s_k + key * KP + d4 * 4) = h;
        if constexpr (NS == 2) *reinterpret_cast<v4*>(s_k + K_EL + key * KP + d4 * 4) = M_::cvt_lo(v, h);
    }
    for (int idx = t; idx < (TK / 4) * D4; idx += NTHR) {
        const int kg = idx / D4, d4 = idx % D4;
        f4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = kg * 4 + j;
            r[j] = f4{0.f, 0.f, 0.f, 0.f};
            if (key < T) r[j] = *reinterpret_cast<const f4*>(base + (long)tok(key) * row3 + a.voff + d4 * 4);
        }
        const f4 c[4] = {{r[0].x, r[1].x, r[2].x, r[3].x}, {r[0].y, r[1].y, r[2].y, r[3].y},
                         {r[0].z, r[1].z, r[2].z, r[3].z}, {r[0].w, r[1].w, r[2].w, r[3].w}};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4 h = M_::cvt(c[q]);
            *reinterpret_cast<v4*>(s_v + (d4 * 4 + q) * VP + kg * 4) = h;
            if constexpr (NS == 2) *reinterpret_cast<v4*>(s_v + V_EL + (d4 * 4 + q) * VP + kg * 4) = M_::cvt_lo(c[q], h);
        }
    }
    }
    // LePE geometry, resolved ONCE per workgroup: byte offset (inside a V^T row) of each of the nine taps of every query slot; a tap
    // outside the window points at the zero column TK (VP = TK + 4), so the per-channel loop below carries no bounds logic
    __shared__ unsigned short s_tap[LEPE ? TK * 9 : 1];
    if constexpr (LEPE) {
        for (int q = t; q < NS * D; q += NTHR) s_v[(q / D) * V_EL + (q % D) * VP + TK] = 0;
        for (int q = t; q < T * 9; q += NTHR) {
            const int slot = q / 9, tap = q - slot * 9;
            const int ty = srow(slot), tx = slot - ty * a.Wsp;
            const int yy = ty + tap / 3 - 1, xx = tx + tap % 3 - 1;
            s_tap[q] = (unsigned short)(((yy >= 0 && yy < a.Hsp && xx >= 0 && xx < a.Wsp) ? yy * a.Wsp + xx : TK) * 2);
        }
    }
    __syncthreads();

    // ---- phase B: each wave owns 16-query tiles ------------------------------------------------------------------------
    slab_t* slab = s_o + wave * 16 * OP;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        const int qt = wave + iq * NW;
        if (qt >= nqt) break;
        // Q fragments (B operand of S^T = K.Q^T): column q = l15, k = d = ks*32 + g*8 + [0,8)
        const int qs = qt * 16 + l15;
        v8 qf[D / 32][NS];
        {
            const gel* qrow = base + (long)tok(qs < T ? qs : 0) * row3;
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) {
                f4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (IO16) {
                    const v8 raw = qraw[iq][ks];                  // zero beyond T
                    if (!a.pre_scale) { qf[ks][0] = raw; continue; }
                    lo4 = f4{(float)raw[0], (float)raw[1], (float)raw[2], (float)raw[3]};
                    hi4 = f4{(float)raw[4], (float)raw[5], (float)raw[6], (float)raw[7]};
                } else if (qs < T) {
                    lo4 = *reinterpret_cast<const f4*>(qrow + ks * 32 + g * 8);
                    hi4 = *reinterpret_cast<const f4*>(qrow + ks * 32 + g * 8 + 4);
                }
                if (a.pre_scale) { lo4 = lo4 * a.scale; hi4 = hi4 * a.scale; }
                const v4 h0 = M_::cvt(lo4), h1 = M_::cvt(hi4);
                qf[ks][0] = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                if constexpr (NS == 2) {
                    const v4 e0 = M_::cvt_lo(lo4, h0), e1 = M_::cvt_lo(hi4, h1);
                    qf[ks][1] = v8{e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                }
            }
        }
        f4 o[D / 16];
        float sum;
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // S^T tiles: lane holds S^T[key = kt*16 + g*4 + r][q = l15]
        f4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f4 acc = zero4;                                       // first MFMA of the chain takes the constant as its C operand
            if (kt < TFULL || kt * 16 < T) {                     // kt < TFULL folds at compile time: no branch, constant C operand
#pragma unroll
                for (int ks = 0; ks < D / 32; ++ks) {
                    v8 kf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        kf[sp] = *reinterpret_cast<const v8*>(s_k + sp * K_EL + (kt * 16 + l15) * KP + ks * 32 + g * 8);
                    acc = mma_step<PREC>(kf, qf[ks], acc);
                }
            }
            s[kt] = acc;
        }
        // softmax over keys (masked beyond T)
        // logits are kept in log2 units: p = 2^(s*scale*log2(e) - max) == exp(s*scale - max) on the v_exp_f32 unit.  The product is
        // rounded BEFORE the maximum is subtracted (not fused into one FMA): the row maximum then maps to exactly 2^0, which the
        // bit-exact index tests rely on (one-hot attention must reproduce V rows exactly).
        const float post = (a.pre_scale ? 1.0f : a.scale) * 1.44269504088896340736f;
        float m = -INFINITY;
        sum = 0.f;
        if constexpr (PREC != 0 && FMA_SM) {
            // fp16 / bf16 operand modes: maximum over the RAW scores (post > 0), then p = 2^(s * post - m * post) with ONE fma per score
            // in front of the v_exp (the separate multiply pass is gone: 56 VALU instructions per query tile).  The row maximum maps to
            // 2^(rounding residue of m * post) = 1 +- 1e-7 instead of exactly 1 -- invisible once P is rounded to 16 bit; the strict
            // mode keeps the exact form for the bit-exact index tests.
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + g * 4 + r;
                    const float v = (kt < TFULL || key < T) ? s[kt][r] : -INFINITY;
                    s[kt][r] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 16, WAVE));
            m = fmaxf(m, __shfl_xor(m, 32, WAVE));
            const float mneg = -(m * post);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], post, mneg));         // fma(-inf, post, .) = -inf -> 0
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + g * 4 + r;
                    const float v = (kt < TFULL || key < T) ? s[kt][r] * post : -INFINITY;      // kt < TFULL folds at compile time
                    s[kt][r] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 16, WAVE));
            m = fmaxf(m, __shfl_xor(m, 32, WAVE));
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kt][r] - m);      // 2^(-inf) = 0 for masked keys
                    s[kt][r] = p;
                    sum += p;
                }
        }
        if constexpr (!(PREC != 0 && FMA_SM)) {
            sum += __shfl_xor(sum, 16, WAVE);
            sum += __shfl_xor(sum, 32, WAVE);
        }
        // O^T = V^T . P^T : A = V^T (row d = l15, k enumerates keys as (tile 2kb, g, r) then (tile 2kb+1, g, r)); B = P^T, the S^T
    // accumulators re-packed in-lane (column q = l15, same key enumeration) -> lane holds O[q = l15][d = nt*16 + g*4 + r]
#pragma unroll
        for (int nt = 0; nt < D / 16; ++nt) o[nt] = zero4;
        // 16-bit operand modes: the row sum comes off the matrix pipe as well -- one more A tile whose rows are all ones gives
        // sum_k P^T[k][q] in every row of its accumulator, i.e. in THIS lane for query l15: 56 VALU adds and two lane-group exchanges
        // per query tile become KT/2 MFMAs (the pipe is 22 % busy), and the normaliser is the sum of the very P values (rounded to
        // 16 bit) that enter the numerator
        f4 osum = zero4;
        typedef typename M_::e el1_t;
        const v8 ones = v8{(el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f, (el1_t)1.0f};
#pragma unroll
        for (int kb = 0; kb < KT / 2; ++kb) {
            if (2 * kb < TFULL || kb * 32 < T) {                  // compile-time true for the full key tiles
                v8 pf[NS];
                {
                    const f4 p0 = s[2 * kb], p1 = s[2 * kb + 1];
                    const v4 h0 = M_::cvt(p0), h1 = M_::cvt(p1);
                    pf[0] = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    if constexpr (NS == 2) {
                        const v4 e0 = M_::cvt_lo(p0, h0), e1 = M_::cvt_lo(p1, h1);
                        pf[1] = v8{e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                    }
                }
#pragma unroll
                for (int nt = 0; nt < D / 16; ++nt) {
                    v8 vf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) {
                        const unsigned short* vr = s_v + sp * V_EL + (nt * 16 + l15) * VP + kb * 32 + g * 4;
                        const v4 a0 = *reinterpret_cast<const v4*>(vr);
                        const v4 a1 = *reinterpret_cast<const v4*>(vr + 16);
                        vf[sp] = v8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    }
                    o[nt] = mma_step<PREC>(vf, pf, o[nt]);      // O^T = V^T . P^T
                }
                if constexpr (PREC != 0 && FMA_SM) osum = M_::mma(ones, pf[0], osum);
            }
        }
        if constexpr (PREC != 0 && FMA_SM) sum = osum.x;
        // normalise.  O was accumulated TRANSPOSED (O^T = V^T . P^T, below): lane (l15, g) holds O[query l15][channel nt*16 + g*4 + r],
        // i.e. the accumulator rows are the rows of the softmax statistics -- the row sum is already in this lane (no shuffles), one
        // reciprocal per lane, and a lane's four values are four consecutive channels of one token: one 8- / 16-byte slab write per
        // tile (the query-major accumulator layout of rounds 1-2 needed 4 shuffles + 4 reciprocals + sixteen 2-byte writes here)
        const float inv = __builtin_amdgcn_rcpf(sum);               // 1 ulp; the result is rounded to 16 bit or scaled once
        // locally-enhanced positional encoding: dw 3x3 over the (Hsp x Wsp) window image of V, zero halo; tap offsets from s_tap
        int off[LEPE ? 9 : 1];
        bool qlive = false;
        if constexpr (LEPE) {
            const int qslot = qt * 16 + l15;
            qlive = qslot < T;
            const unsigned short* tp = s_tap + (qlive ? qslot : 0) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) off[i] = tp[i];
        }
#pragma unroll
        for (int nt = 0; nt < D / 16; ++nt) {
            f4 val = o[nt] * inv;
            if constexpr (LEPE) {
                if (qlive) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = nt * 16 + g * 4 + r;                        // channel inside this head
                        const float* lwp = s_lw + d * 10;
                        const char* vrow = reinterpret_cast<const char*>(s_v + d * VP);
                        float acc = lwp[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            float vv = (float)(*reinterpret_cast<const el*>(vrow + off[i]));
                            if constexpr (NS == 2) vv += (float)(*reinterpret_cast<const el*>(vrow + V_EL * 2 + off[i]));
                            acc = __builtin_fmaf(lwp[i], vv, acc);
                        }
                        val[r] += acc;
                    }
                }
            }
            if constexpr (IO16) *reinterpret_cast<v4*>(slab + l15 * OP + nt * 16 + g * 4) = M_::cvt(val);
            else *reinterpret_cast<f4*>(slab + l15 * OP + nt * 16 + g * 4) = val;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // row-contiguous stores: 16 bytes per lane
        if constexpr (IO16) {
            constexpr int LPR = D / 8, RPI = 64 / LPR;
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c8 = (lane % LPR) * 8;
                const int qslot = qt * 16 + r;
                if (qslot < T)
                    *reinterpret_cast<v8*>(static_cast<gel*>(a.out) + ((long)b * a.L + tok(qslot)) * a.Ctot + och0 + c8) =
                        *reinterpret_cast<const v8*>(slab + r * OP + c8);
            }
        } else {
            constexpr int LPR = D / 4, RPI = 64 / LPR;
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c4 = (lane % LPR) * 4;
                const int qslot = qt * 16 + r;
                if (qslot < T)
                    *reinterpret_cast<f4*>(static_cast<gel*>(a.out) + ((long)b * a.L + tok(qslot)) * a.Ctot + och0 + c4) =
                        *reinterpret_cast<const f4*>(slab + r * OP + c4);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

template <int D, bool LEPE, bool IO16>
int launch_attn(const AttnArgs& a, int B, int precision, hipStream_t st, const AttnArgs* other = nullptr) {
    AttnPair pr{};
    pr.a0 = a;
    pr.split = B * a.nwin * a.heads;
    int grid = pr.split;
    if (other) {
        pr.a1 = *other;
        grid += B * other->nwin * other->heads;
    }
    if (IO16 && precision == MI355_PREC_STRICT)
        return mi355::fail(MI355_EINVAL, "16-bit activation I/O needs precision 1 (fp16) or 2 (bf16)");
    MI355_TRACE(st, "win_attn_kernel<d=%d%s%s> B=%d windows=%d heads=%d tokens=%d", D, LEPE ? ",lepe" : "", IO16 ? ",io16" : "", B, a.nwin, a.heads, a.T);
#define GO(P, KT_, NW_) win_attn_kernel<P, D, KT_, LEPE, (IO16 && P != 0), NW_><<<grid, NW_ * 64, 0, st>>>(pr)
#define GO_OCC(P, KT_, NW_, OCC_) win_attn_kernel<P, D, KT_, LEPE, (IO16 && P != 0), NW_, ((IO16 && P != 0 && D == 32) ? OCC_ : 1)><<<grid, NW_ * 64, 0, st>>>(pr)
#define GO_FULL(P, KT_, NW_, OCC_, TF_) \
    win_attn_kernel<P, D, KT_, LEPE, (IO16 && P != 0), NW_, ((IO16 && P != 0 && D == 32) ? OCC_ : 1), TF_><<<grid, NW_ * 64, 0, st>>>(pr)
    const bool hot = IO16 && (!other || other->T == a.T);        // 16-bit I/O shapes of the models: compile-time count of full key tiles
#define BYKT(P)                                          \
    do {                                                 \
        if (a.T <= 64) { if (hot && P != 0 && a.T >= 48) GO_FULL(P, 4, 4, 6, 3); else GO_OCC(P, 4, 4, 6); }          \
        else if (a.T <= 128) { if (hot && P != 0 && a.T >= 96) GO_FULL(P, 8, 4, 5, 6); else GO_OCC(P, 8, 4, 5); }    \
        else if (IO16 && P != 0) {                                                                                    \
            /* 193 .. 208 tokens = 13 query tiles: on 8 waves five waves carry two tiles and three carry one (13 / 16), on 7 waves */ \
            /* six carry two and one carries one (13 / 14): option "attn_nw" (A/B switch, round 5) */                    \
            if (a.T >= 192 && a.T <= 208 && mi355::opt_attn_nw() == 7) GO_FULL(P, 14, 7, 1, 12);                          \
            else if (a.T >= 192) GO_FULL(P, 14, 8, 1, 12); else GO(P, 14, 8); }                                          \
        else GO(P, 14, 4);                               \
    } while (0)
    switch (precision) {
        case MI355_PREC_STRICT: BYKT(0); break;
        case MI355_PREC_FP16:   BYKT(1); break;
        case MI355_PREC_BF16:   BYKT(2); break;
        default: return mi355::fail(MI355_EINVAL, "precision must be 0, 1 or 2 (got %d)", precision);
    }
#undef BYKT
#undef GO_FULL
#undef GO_OCC
#undef GO
    return MI355_OK;
}

}  // namespace

static int sdpa_common(const void* qkv, void* out, int B, int N, int heads, int d, float scale, int precision, bool io16,
                       hipStream_t st) {
    AttnArgs a{};
    a.qkv = qkv; a.out = out; a.L = N; a.Ctot = heads * d; a.c0 = 0; a.heads = heads;
    a.koff = a.Ctot; a.voff = 2 * a.Ctot; a.oc0 = 0;
    a.reso = N; a.Hsp = 1; a.Wsp = N; a.nWx = 1; a.nwin = 1; a.T = N; a.scale = scale; a.pre_scale = 0;
    a.wsp_magic = (unsigned)(((1ull << 32) + (unsigned)N - 1) / (unsigned)N);
    if (d == 64) return io16 ? launch_attn<64, false, true>(a, B, precision, st) : launch_attn<64, false, false>(a, B, precision, st);
    return io16 ? launch_attn<32, false, true>(a, B, precision, st) : launch_attn<32, false, false>(a, B, precision, st);
}

static AttnArgs lepe_args(const void* qkv, const float* getv_w, const float* getv_b, void* out, int reso, int Ctot, int c0, int heads,
                          int Hsp, int Wsp, float scale) {
    AttnArgs a{};
    a.qkv = qkv; a.out = out; a.lepe_w = getv_w; a.lepe_b = getv_b;
    a.L = reso * reso; a.Ctot = Ctot; a.c0 = c0; a.heads = heads;
    a.koff = Ctot; a.voff = 2 * Ctot; a.oc0 = c0;              // reference column order of the qkv projection: [q | k | v]
    a.reso = reso; a.Hsp = Hsp; a.Wsp = Wsp; a.nWx = reso / Wsp; a.nwin = (reso / Hsp) * (reso / Wsp); a.T = Hsp * Wsp;
    a.scale = scale; a.pre_scale = 1;
    a.wsp_magic = (unsigned)(((1ull << 32) + (unsigned)Wsp - 1) / (unsigned)Wsp);
    return a;
}

static int lepe_common(const void* qkv, const float* getv_w, const float* getv_b, void* out, int B, int reso, int Ctot, int c0, int Cb,
                       int heads, int Hsp, int Wsp, float scale, int precision, bool io16, hipStream_t st) {
    (void)Cb;
    const AttnArgs a = lepe_args(qkv, getv_w, getv_b, out, reso, Ctot, c0, heads, Hsp, Wsp, scale);
    return io16 ? launch_attn<32, true, true>(a, B, precision, st) : launch_attn<32, true, false>(a, B, precision, st);
}

#define SDPA_CHECKS(fn)                                                                                                   \
    MI355_CHECK_ARG(qkv && out && B > 0 && N > 0 && heads > 0);                                                           \
    if (!(d == 32 || d == 64)) return mi355::fail(MI355_EUNSUPPORTED, fn ": head dim %d (built: 32, 64)", d);            \
    if (N > 224) return mi355::fail(MI355_EUNSUPPORTED, fn ": sequence length %d > 224 (single-pass softmax core)", N);   \
    MI355_CHECK_ARG(aligned16(qkv) && aligned16(out))

#define LEPE_CHECKS(fn)                                                                                                    \
    MI355_CHECK_ARG(qkv && getv_w && getv_b && out);                                                                       \
    MI355_CHECK_ARG(B > 0 && reso > 0 && Ctot > 0 && c0 >= 0 && Cb > 0 && c0 + Cb <= Ctot && heads > 0 && Cb % heads == 0); \
    MI355_CHECK_ARG(Hsp > 0 && Wsp > 0 && reso % Hsp == 0 && reso % Wsp == 0);                                             \
    if (Cb / heads != 32) return mi355::fail(MI355_EUNSUPPORTED, fn ": head dim %d (built: 32)", Cb / heads);             \
    if (Hsp * Wsp > 224) return mi355::fail(MI355_EUNSUPPORTED, fn ": %d tokens per stripe window > 224", Hsp * Wsp);     \
    MI355_CHECK_ARG((Ctot & 7) == 0 && (c0 & 7) == 0 && aligned16(qkv) && aligned16(out))

extern "C" {

int mi355_sdpa_fwd(const float* qkv, float* out, int B, int N, int heads, int d, float scale, int precision,
                   mi355_stream_t stream) {
    SDPA_CHECKS("mi355_sdpa_fwd");
    int rc = sdpa_common(qkv, out, B, N, heads, d, scale, precision, false, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_sdpa16_fwd(const void* qkv, void* out, int B, int N, int heads, int d, float scale, int precision,
                     mi355_stream_t stream) {
    SDPA_CHECKS("mi355_sdpa16_fwd");
    int rc = sdpa_common(qkv, out, B, N, heads, d, scale, precision, true, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_cswin_lepe_attn_fwd(const float* qkv, const float* getv_w, const float* getv_b, float* out, int B, int reso, int Ctot,
                              int c0, int Cb, int heads, int Hsp, int Wsp, float scale, int precision, mi355_stream_t stream) {
    LEPE_CHECKS("mi355_cswin_lepe_attn_fwd");
    int rc = lepe_common(qkv, getv_w, getv_b, out, B, reso, Ctot, c0, Cb, heads, Hsp, Wsp, scale, precision, false,
                         static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_cswin_lepe_attn16_pair_fwd(const void* qkv, const float* getv_w0, const float* getv_b0, const float* getv_w1, const float* getv_b1,
                                     void* out, int B, int reso, int Ctot, int heads, int split, float scale, int precision,
                                     mi355_stream_t stream) {
    MI355_CHECK_ARG(qkv && getv_w0 && getv_b0 && getv_w1 && getv_b1 && out);
    MI355_CHECK_ARG(B > 0 && reso > 0 && Ctot > 0 && heads > 0 && (Ctot / 2) % heads == 0 && split > 0 && reso % split == 0);
    if (Ctot / 2 / heads != 32) return mi355::fail(MI355_EUNSUPPORTED, "mi355_cswin_lepe_attn16_pair_fwd: head dim %d (built: 32)", Ctot / 2 / heads);
    if (reso * split > 224) return mi355::fail(MI355_EUNSUPPORTED, "mi355_cswin_lepe_attn16_pair_fwd: %d tokens per stripe window > 224", reso * split);
    MI355_CHECK_ARG((Ctot & 15) == 0 && aligned16(qkv) && aligned16(out));
    // branch 0: idx 0 of cswin.py:62-67 (H_sp = resolution, W_sp = split) on channels [0, C/2); branch 1: idx 1 on [C/2, C)
    const AttnArgs a0 = lepe_args(qkv, getv_w0, getv_b0, out, reso, Ctot, 0, heads, reso, split, scale);
    const AttnArgs a1 = lepe_args(qkv, getv_w1, getv_b1, out, reso, Ctot, Ctot / 2, heads, split, reso, scale);
    int rc = launch_attn<32, true, true>(a0, B, precision, static_cast<hipStream_t>(stream), &a1);
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_cswin_lepe_attn16_fwd(const void* qkv, const float* getv_w, const float* getv_b, void* out, int B, int reso, int Ctot,
                                int c0, int Cb, int heads, int Hsp, int Wsp, float scale, int precision, mi355_stream_t stream) {
    LEPE_CHECKS("mi355_cswin_lepe_attn16_fwd");
    int rc = lepe_common(qkv, getv_w, getv_b, out, B, reso, Ctot, c0, Cb, heads, Hsp, Wsp, scale, precision, true,
                         static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
