// cswin_fused.hip -- first half of a CSWinBlock for the narrow stages (C = 64 / 128: CSWin-T stages 1-2) in ONE kernel:
//
//     LayerNorm(x) -> qkv Linear -> stripe-window attention with LePE        cswin.py:180-190 with LePEAttention.forward :101-127
//
// Unfused, the block half moved 4.5-6x its algorithmic bytes (profiles/r02_pmc_blocks.jsonl: 2.59 GB per stage-1 block): the qkv
// tensor (3 x the activation, 16 bit) was written and read back, and the attention workgroups were one-shot -- one memory round
// trip + one barrier + a few MFMAs per (image, window, head).  Here a PERSISTENT workgroup (4 waves) owns one (branch, head) unit =
// 32 q + 32 k + 32 v output features, keeps that 96 x C slice of the (LayerNorm-folded) projection in registers (MFMA fragments), and walks the stripe
// windows of its unit: the fp32 rows of a window's T <= 64 tokens are normalised in registers (4 lanes per token), parked in LDS
// in MFMA operand format, projected (6 feature tiles x C/32 MFMAs per wave: q and k as W.Xn^T so that a lane holds 4 consecutive
// features of a token, v as Xn.W^T so that it holds 4 consecutive tokens of a feature = the V^T layout the PV product reads),
// and attended like attn.hip (S^T = K.Q^T, softmax in registers, P re-packed in-lane).  LePE (depth-wise 3x3 over the window image of
// v, zero padded at the window border) is evaluated in the STORE layout -- lane = (query row, 8 channels) -- from a second,
// token-major copy of v: the window geometry is the same for every window of a unit, so a lane's nine neighbour rows and its
// 72 tap weights are kernel-lifetime constants in registers and a window costs it nine 16-byte LDS reads + 72 FMAs (the
// accumulator-layout version of attn.hip -- 72 two-byte reads with their address arithmetic per lane -- was 42 % of this kernel).
// The next window's x rows are in flight while the current one is computed; two barriers per window.  x is read once per unit
// (2x per block at stage 1, 4x at stage 2 -- from the Infinity Cache after the first), qkv never exists in HBM, ctx leaves as
// whole 64-byte head rows in 16 bit.
#include "common.h"
#include "mma.h"

namespace {

struct StripeArgs {
    const float* x;            // (B, L, C) fp32
    const void* w;             // (3C, C) 16-bit, LayerNorm affine folded in (W' = W diag(ln_w))
    const float* b;            // (3C) fp32, folded (b' = b + W ln_b)
    const float* lw[2];        // get_v.weight of the two branches: (C/2, 1, 3, 3)
    const float* lb[2];        // get_v.bias: (C/2)
    void* ctx;                 // (B, L, C) 16-bit
    int B, reso, split, hb;    // hb: heads per branch
    float scale, eps;
    int win_per_img;           // windows per image and branch (reso / split)
    unsigned* ovf;             // fp16 range word (common.h, code 4): q / k / v and ctx are 16-bit conversions of unbounded products; null for bf16
};


// One LePE tap on eight channels: r += w * (float)nb.  fp16: v_fma_mix_f32 reads the packed 16-bit neighbour values as they are
// (hipcc otherwise converts all eight first -- eight v_cvt_f32_f16 in front of four v_pk_fma_f32; the mixed form needs no conversions); same arithmetic.
template <int PREC, typename V8>
__device__ __forceinline__ void lepe_tap(f4& r0, f4& r1, const V8 nb, const f4 w0, const f4 w1) {
    if constexpr (PREC == 1) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t p = __builtin_bit_cast(u32x4_t, nb);
#define MI355_MIX(acc, pk, wt)                                                                                             \
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.x) : "v"(pk.x), "v"(wt.x));           \
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.y) : "v"(pk.x), "v"(wt.y));           \
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.z) : "v"(pk.y), "v"(wt.z));           \
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.w) : "v"(pk.y), "v"(wt.w));
        struct { unsigned x, y; } lo{p.x, p.y}, hi{p.z, p.w};
        MI355_MIX(r0, lo, w0)
        MI355_MIX(r1, hi, w1)
#undef MI355_MIX
    } else {
        r0 = __builtin_elementwise_fma(w0, f4{(float)nb[0], (float)nb[1], (float)nb[2], (float)nb[3]}, r0);
        r1 = __builtin_elementwise_fma(w1, f4{(float)nb[4], (float)nb[5], (float)nb[6], (float)nb[7]}, r1);
    }
}

template <int PREC, int C, int TFULL = 0, bool L1D = false, int ABL = 0>
__global__ __launch_bounds__(C * 4, (C == 64 ? 3 : 2)) void cswin_stripe_kernel(const StripeArgs a) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    static_assert(Mma<PREC>::NSPLIT == 1, "16-bit operand modes only");
    constexpr int D = 32, TK = 64, KT = 4;
    constexpr int NH = C / 64;                 // heads per branch = head groups of four waves in this workgroup
    constexpr int NW = 4 * NH, NTHR = 64 * NW;
    constexpr bool LWREG = true;
    constexpr int KS = C / 32;                 // k-steps of the projection
    constexpr int WP = C + 8;                  // pitch of s_w / s_xn rows (elements)
    constexpr int QP = D + 8;                  // pitch of s_q / s_k rows
    constexpr int VP = TK + 4;                 // pitch of s_v rows (V^T: [d][key]); column TK is the zero column of the LePE taps
    constexpr int OP = D + 4;                  // slab pitch (floats)
    constexpr int PT = C / (4 * NH);           // floats of a token row per staging thread (4 NH threads per token): 16
    __shared__ __attribute__((aligned(16))) unsigned short s_xn[TK * WP];
    __shared__ __attribute__((aligned(16))) unsigned short s_q_[NH][TK * QP];
    __shared__ __attribute__((aligned(16))) unsigned short s_k_[NH][TK * QP];
    __shared__ __attribute__((aligned(16))) unsigned short s_v_[NH][D * VP];
    __shared__ __attribute__((aligned(16))) unsigned short s_vt_[NH][(TK + 1) * QP];   // v token-major ([key][d]); row TK is the zero row of the LePE taps
    __shared__ __attribute__((aligned(16))) float s_o[NW * 16 * OP];
    __shared__ __attribute__((aligned(16))) float s_bias_[NH][96];
    __shared__ __attribute__((aligned(16))) float s_lwt_[NH][10 * D];   // LePE taps, tap-major: [tap][channel]; row 9 = bias

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    // ---- unit of this workgroup: a BRANCH; the first half of the grid walks the windows of branch 0, the second half those of
    //      branch 1.  The NH heads of the branch are groups of four waves that share the LayerNorm'd window in s_xn ------------------
    const int per_unit = gridDim.x / 2;                            // launcher: gridDim.x is even
    const int br = blockIdx.x / per_unit, slot = blockIdx.x % per_unit;
    const int head = __builtin_amdgcn_readfirstlane(wave >> 2), wq = wave & 3;   // head of this wave, its token tile inside the window
    unsigned short* const s_q = s_q_[head];
    unsigned short* const s_k = s_k_[head];
    unsigned short* const s_v = s_v_[head];
    unsigned short* const s_vt = s_vt_[head];
    float* const s_bias = s_bias_[head];
    float* const s_lwt = s_lwt_[head];
    const int Hsp = br == 0 ? a.reso : a.split, Wsp = br == 0 ? a.split : a.reso, nWx = a.reso / Wsp;
    const int T = a.reso * a.split;                                // tokens per window (<= 64)
    const int L = a.reso * a.reso;
    const int ch0 = br * (C / 2) + head * D;                       // first channel of this unit inside a C-wide row
    const el* wg = static_cast<const el*>(a.w);

    // ---- once per workgroup: weight slice, bias slice, LePE taps / tap offsets ----------------------------------------------------
    // the unit's 96 x C weight slice lives in REGISTERS as MFMA fragments (6 feature tiles x C/32 k-steps x 4 VGPRs): every wave
    // needs all of it for every window, and both orientations of the product take the same fragment
    v8 wfr[6][KS];
#pragma unroll
    for (int ft = 0; ft < 6; ++ft) {
        const int f = ft * 16 + l15;
        const int grow = (f >> 5) * C + ch0 + (f & 31);            // q | k | v rows of the (3C, C) projection
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wfr[ft][ks] = *reinterpret_cast<const v8*>(wg + (long)grow * C + ks * 32 + g * 8);
    }
    if (t < 96 * NH) {
        const int hh = t / 96, tt = t - hh * 96;
        s_bias_[hh][tt] = a.b[(tt >> 5) * C + br * (C / 2) + hh * D + (tt & 31)];
    }
    for (int q = t; q < NH * QP; q += NTHR) s_vt_[q / QP][TK * QP + q % QP] = 0;
    // LePE constants of this lane: it stores query row wq*16 + lane/4, channels (lane & 3)*8 + [0,8) of every window
    const int sr = lane >> 2, sc8 = (lane & 3) * 8;
    const int myslot = wq * 16 + sr;
    // L1D (split == 1: the stripes are one token wide, CSWin stage 1): six of the nine taps fall outside the window for EVERY token,
    // the other three are the slots before / at / after the token in both branches (branch 0: taps (dy, 0) = 1, 4, 7; branch 1:
    // taps (0, dx) = 3, 4, 5).  Three LDS rows instead of nine and the 24 tap weights of a lane live in registers.
    constexpr int NTAP = L1D ? 3 : 9;
    int tapoff[NTAP];                                              // element offset of the neighbour's row in s_vt (zero row outside the window)
    f4 lw0[L1D && LWREG ? 3 : 1], lw1[L1D && LWREG ? 3 : 1];       // L1D: tap weights of channels sc8 + [0,4) / + [4,8)
    int tap0 = 0, tapstep = 1;
    if constexpr (L1D) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sl = myslot + j - 1;
            tapoff[j] = ((myslot < T && sl >= 0 && sl < T) ? sl : TK) * QP + sc8;
            const int tap = br == 0 ? 1 + 3 * j : 3 + j;
            if constexpr (LWREG) {
                const float* wsrc = a.lw[br] + (long)(head * D + sc8) * 9 + tap;
                lw0[j] = f4{wsrc[0], wsrc[9], wsrc[18], wsrc[27]};
                lw1[j] = f4{wsrc[36], wsrc[45], wsrc[54], wsrc[63]};
            }
        }
        tap0 = br == 0 ? 1 : 3; tapstep = br == 0 ? 3 : 1;
    } else {
        const int ty = myslot / Wsp, tx = myslot - ty * Wsp;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = ty + tap / 3 - 1, xx = tx + tap % 3 - 1;
            tapoff[tap] = ((myslot < T && yy >= 0 && yy < Hsp && xx >= 0 && xx < Wsp) ? yy * Wsp + xx : TK) * QP + sc8;
        }
    }
    for (int q = t; q < NH * 10 * D; q += NTHR) {
        const int hh = q / (10 * D), qq = q - hh * 10 * D, tap = qq / D, c = qq - tap * D;
        s_lwt_[hh][qq] = tap < 9 ? a.lw[br][(long)(hh * D + c) * 9 + tap] : a.lb[br][hh * D + c];
    }
    // rows of s_xn beyond T stay zero for the whole kernel (their q / k / v are bias-only and masked / never stored)
    for (int i = t; i < TK * WP / 2; i += NTHR) reinterpret_cast<unsigned int*>(s_xn)[i] = 0u;

    // window slot -> token: token = origin(window) + offset(slot); branch 0 windows are columns of width `split` (origin = win * split),
    // branch 1 windows are rows of height `split` (origin = win * split * reso).  No divisions inside the window loop.
    auto slot_off = [&](int sl) { const int r = sl / Wsp; return r * a.reso + (sl - r * Wsp); };
    const int org_step = br == 0 ? a.split : a.split * a.reso;
    (void)nWx;

    // ---- x staging: thread -> (token slot, 16-float piece of the row): 4 NH threads per token -------------------------------------------
    const int xs = t / (4 * NH), xq = t % (4 * NH);
    f4 xr[PT / 4];
    const int nwin_total = a.B * a.win_per_img;                    // launcher: < 2^31
    const int xoff = slot_off(xs < T ? xs : 0), soff = slot_off(myslot < T ? myslot : 0);
    const int step_b = per_unit / a.win_per_img, step_w = per_unit % a.win_per_img;
    auto load_x = [&](int b, int win) {
        const float* p = a.x + ((long)b * L + win * org_step + xoff) * C + xq * PT;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) xr[i] = *reinterpret_cast<const f4*>(p + i * 4);
    };
    auto commit_x = [&]() {                                        // LayerNorm (affine folded into the projection) -> s_xn, 16 bit
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) s += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
        s += __shfl_xor(s, 1, WAVE);
        s += __shfl_xor(s, 2, WAVE);
        if (NH == 2) s += __shfl_xor(s, 4, WAVE);
        const float mean = s * (1.0f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) {
            const f4 d = xr[i] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        q += __shfl_xor(q, 1, WAVE);
        q += __shfl_xor(q, 2, WAVE);
        if (NH == 2) q += __shfl_xor(q, 4, WAVE);
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)C) + a.eps);
        if (xs < T) {
#pragma unroll
            for (int j = 0; j < PT / 8; ++j) {
                const f4 a0 = (xr[2 * j] - mean) * rstd, a1 = (xr[2 * j + 1] - mean) * rstd;
                const v4 h0 = M_::cvt(a0), h1 = M_::cvt(a1);
                *reinterpret_cast<v8*>(s_xn + xs * WP + xq * PT + j * 8) = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            }
        }
    };

    int wi = slot;
    int b = slot / a.win_per_img, win = slot % a.win_per_img;
    if (wi < nwin_total) load_x(b, win);
    __syncthreads();                                               // weights / tables / zeroed s_xn visible
    const float L2E = 1.44269504088896340736f;
    float rgmax = 0.f;
    float* slab = s_o + wave * 16 * OP;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (; wi < nwin_total; wi += per_unit) {
        commit_x();
        __syncthreads();                                           // (1) s_xn complete; everybody is done with the previous window's q / k / v
        int nb = b + step_b, nwin = win + step_w;
        if (nwin >= a.win_per_img) { nwin -= a.win_per_img; ++nb; }
        if (wi + per_unit < nwin_total) load_x(nb, nwin);          // next window's rows fly during the math below

        // ---- projection of this wave's 16 tokens: 6 feature tiles -------------------------------------------------------------------
        if (!(ABL & 4)) {
            v8 xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const v8*>(s_xn + (wq * 16 + l15) * WP + ks * 32 + g * 8);
#pragma unroll
            for (int ft = 0; ft < 6; ++ft) {
                // the bias is the initial value of the accumulator (no separate add): q / k tiles hold features g*4 + r of token l15,
                // v tiles feature l15 of tokens g*4 + r
                const f4 brow = *reinterpret_cast<const f4*>(s_bias + ft * 16 + g * 4);
                const float bcol = s_bias[ft * 16 + l15];
                f4 acc = ft < 4 ? brow : f4{bcol, bcol, bcol, bcol};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const v8 wf = wfr[ft][ks];
                    if (ft < 4) acc = M_::mma(wf, xf[ks], acc);    // q, k: rows = features, columns = tokens
                    else        acc = M_::mma(xf[ks], wf, acc);    // v:    rows = tokens,   columns = features
                }
                if (ft < 4) {
                    f4 v = acc;
                    if (ft < 2) v = v * a.scale;                   // cswin.py:116: q * scale before the product
                    if constexpr (PREC == 1) rgmax = rg_max3abs4(rgmax, v);
                    const v4 h = M_::cvt(v);
                    unsigned short* dst = (ft < 2 ? s_q : s_k) + (wq * 16 + l15) * QP + (ft & 1) * 16 + g * 4;
                    *reinterpret_cast<v4*>(dst) = h;
                } else {
                    if constexpr (PREC == 1) rgmax = rg_max3abs4(rgmax, acc);       // (acc2 below holds the same values transposed)
                    const v4 h = M_::cvt(acc);
                    *reinterpret_cast<v4*>(s_v + ((ft - 4) * 16 + l15) * VP + wq * 16 + g * 4) = h;
                    // the same tile token-major for LePE (second orientation of the product: the matrix pipe has room)
                    f4 acc2 = brow;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc2 = M_::mma(wfr[ft][ks], xf[ks], acc2);
                    const v4 h2 = M_::cvt(acc2);
                    *reinterpret_cast<v4*>(s_vt + (wq * 16 + l15) * QP + (ft - 4) * 16 + g * 4) = h2;
                }
            }
        }
        __syncthreads();                                           // (2) q / k / v of all 64 slots in LDS

        // ---- attention of this wave's query tile (attn.hip phase B, D = 32, KT = 4) --------------------------------------------------
        const int qt = wq;
        if (qt * 16 < T && !(ABL & 2)) {
            const v8 qf = *reinterpret_cast<const v8*>(s_q + (qt * 16 + l15) * QP + g * 8);
            f4 s[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                s[kt] = zero4;
                if (kt * 16 < T) {
                    const v8 kf = *reinterpret_cast<const v8*>(s_k + (kt * 16 + l15) * QP + g * 8);
                    s[kt] = M_::mma(kf, qf, s[kt]);
                }
            }
            // softmax over the keys of query l15 (rows g*4 + r of the four key tiles): the maximum is taken over the raw scores and
            // log2(e) rides in the FMA that forms the exponent (attn.hip)
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + g * 4 + r;
                    const float v = (kt < TFULL || key < T) ? s[kt][r] : -INFINITY;            // kt < TFULL folds at compile time
                    s[kt][r] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 16, WAVE));
            m = fmaxf(m, __shfl_xor(m, 32, WAVE));
            const float nm = -(m * L2E);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], L2E, nm));
            // O^T = V^T P^T: the accumulator of a lane holds features nt*16 + g*4 + r of query l15 -- the query its softmax
            // statistics belong to, so the normaliser needs no exchange and a lane's four features leave as one 16-byte write
            // the row sum comes off the matrix pipe too (an A tile of ones: every accumulator row = sum_k P^T[k][q], attn.hip)
            f4 o[2] = {zero4, zero4}, osum = zero4;
            const v8 ones = v8{(el)1.0f, (el)1.0f, (el)1.0f, (el)1.0f, (el)1.0f, (el)1.0f, (el)1.0f, (el)1.0f};
#pragma unroll
            for (int kb = 0; kb < KT / 2; ++kb) {
                if (kb * 32 < T) {
                    const v4 h0 = M_::cvt(s[2 * kb]), h1 = M_::cvt(s[2 * kb + 1]);
                    const v8 pf = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const unsigned short* vr = s_v + (nt * 16 + l15) * VP + kb * 32 + g * 4;
                        const v4 a0 = *reinterpret_cast<const v4*>(vr), a1 = *reinterpret_cast<const v4*>(vr + 16);
                        o[nt] = M_::mma(v8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, pf, o[nt]);
                    }
                    osum = M_::mma(ones, pf, osum);
                }
            }
            const float inv = __builtin_amdgcn_rcpf(osum.x);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f4*>(slab + l15 * OP + nt * 16 + g * 4) = o[nt] * inv;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (myslot < T) {                                       // 64-byte head rows: 4 lanes per row, 16 rows per instruction
                f4 r0 = *reinterpret_cast<const f4*>(slab + sr * OP + sc8), r1 = *reinterpret_cast<const f4*>(slab + sr * OP + sc8 + 4);
                if (!(ABL & 1)) {
                    // accumulate straight onto the attention output: one mixed-precision FMA per (tap, channel) (v_fma_mix_f32 takes the
                    // 16-bit neighbour value as is)
                    const f4 bz0 = *reinterpret_cast<const f4*>(s_lwt + 9 * D + sc8), bz1 = *reinterpret_cast<const f4*>(s_lwt + 9 * D + sc8 + 4);
                    r0 = r0 + bz0;
                    r1 = r1 + bz1;
#pragma unroll
                    for (int tap = 0; tap < NTAP; ++tap) {
                        const v8 nb = *reinterpret_cast<const v8*>(s_vt + tapoff[tap]);
                        f4 w0, w1;
                        if constexpr (L1D && LWREG) { w0 = lw0[tap]; w1 = lw1[tap]; }
                        else if constexpr (L1D) {
                            const float* wr = s_lwt + (tap0 + tap * tapstep) * D + sc8;
                            w0 = *reinterpret_cast<const f4*>(wr); w1 = *reinterpret_cast<const f4*>(wr + 4);
                        } else { w0 = *reinterpret_cast<const f4*>(s_lwt + tap * D + sc8); w1 = *reinterpret_cast<const f4*>(s_lwt + tap * D + sc8 + 4); }
                        lepe_tap<PREC>(r0, r1, nb, w0, w1);
                    }
                }
                if constexpr (PREC == 1) rgmax = rg_max3abs4(rg_max3abs4(rgmax, r0), r1);
                const v4 h0 = M_::cvt(r0), h1 = M_::cvt(r1);
                *reinterpret_cast<v8*>(static_cast<el*>(a.ctx) + ((long)b * L + win * org_step + soff) * C + ch0 + sc8) =
                    v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        b = nb; win = nwin;
    }
    if constexpr (PREC == 1) rg_report_f(rgmax, a.ovf, 4u);
}

}  // namespace

extern "C" int mi355_cswin_stripe_attn_fwd(const float* x, const void* wqkv16, const float* bqkv, const float* getv_w0, const float* getv_b0,
                                           const float* getv_w1, const float* getv_b1, void* ctx16, int B, int reso, int C,
                                           int heads_per_branch, int split, float scale, float eps, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && wqkv16 && bqkv && getv_w0 && getv_b0 && getv_w1 && getv_b1 && ctx16);
    MI355_CHECK_ARG(B > 0 && reso > 0 && split > 0 && reso % split == 0 && heads_per_branch > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (!(C == 64 || C == 128) || heads_per_branch * 64 != C || reso * split > 64)
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_cswin_stripe_attn_fwd: built for C = 64 / 128, head width 32, <= 64 tokens per stripe "
                           "(C = %d, heads per branch = %d, tokens = %d)", C, heads_per_branch, reso * split);
    MI355_CHECK_ARG(aligned16(x) && aligned16(wqkv16) && aligned16(ctx16) && aligned16(bqkv));
    StripeArgs a{};
    a.x = x; a.w = wqkv16; a.b = bqkv; a.lw[0] = getv_w0; a.lb[0] = getv_b0; a.lw[1] = getv_w1; a.lb[1] = getv_b1; a.ctx = ctx16;
    a.B = B; a.reso = reso; a.split = split; a.hb = heads_per_branch; a.scale = scale; a.eps = eps; a.win_per_img = reso / split;
    const long nwin = (long)B * a.win_per_img;                    // windows per branch
    if (nwin >= (1L << 30)) return mi355::fail(MI355_EUNSUPPORTED, "mi355_cswin_stripe_attn_fwd: too many windows");
    // C = 64: 4-wave workgroups, three per CU; C = 128: 8-wave workgroups (two heads share the LayerNorm'd window), one per CU
    long per_unit = (long)mi355::resident_slots(C == 64 ? 3 : 1) / 2;
    if (per_unit > nwin) per_unit = nwin;
    if (per_unit < 1) per_unit = 1;
    const int grid = (int)(per_unit * 2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.ovf = precision == MI355_PREC_FP16 ? mi355::range_word(st) : nullptr;
    const bool t3 = reso * split >= 48;                           // the model shapes (56 tokens): three key tiles need no validity mask
    MI355_TRACE(st, "cswin_stripe_kernel<C=%d> B=%d reso=%d split=%d", C, B, reso, split);
#define GO(P_, C_)                                                                       \
    do {                                                                                 \
        if (split == 1 && t3) cswin_stripe_kernel<P_, C_, 3, true><<<grid, C_ * 4, 0, st>>>(a);    \
        else if (split == 1)  cswin_stripe_kernel<P_, C_, 0, true><<<grid, C_ * 4, 0, st>>>(a);    \
        else if (t3)          cswin_stripe_kernel<P_, C_, 3, false><<<grid, C_ * 4, 0, st>>>(a);   \
        else                  cswin_stripe_kernel<P_, C_, 0, false><<<grid, C_ * 4, 0, st>>>(a);   \
    } while (0)
    if (C == 64) { if (precision == MI355_PREC_FP16) GO(1, 64); else GO(2, 64); }
    else         { if (precision == MI355_PREC_FP16) GO(1, 128); else GO(2, 128); }
#undef GO
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}
