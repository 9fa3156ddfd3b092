// double_attn_small.hip -- A2-Net DoubleAttention forward (double_attention.py:32-48) in ONE kernel, one pass over the image, for the
// shape class of the reference's own smoke test / README: c_m = c_n = 32, C = 64, H*W <= 1024 (x = (B, 64, 32, 32)).  (C = 128 was
// instantiated and needs 256 registers + 668 B of scratch per lane with this register-resident weight scheme: left to the general path.)
//
//   A = WA X + bA,  Bm = softmax_HW(WB X + bB),  V = softmax_c(WV X + bV),  G = A Bm^T,  y = WP (G V) + bP = (WP G) V + bP
//
// Algorithmic bytes = read x once + write y once = 2 C HW 4 B per image (512 KB at the README shape); FLOPs 2 HW (3.32.C + 32.32 + C.32).
// The general pipeline (double_attn.hip) ran this shape in 7 launches at 0.09 of the HBM roofline (0.18 ms for 134 MB); the two-pass
// kernels of round 2 (double_attn_fused.hip) are built for c = 128.  Here the whole image stays on the CU:
//
//   one 8-wave workgroup per image.  A wave walks its own 32-pixel groups (HW / 32 groups, round-robin over the waves):
//     * the group's C x 32 fp32 slab arrives by plain 16-byte loads (128-byte runs per channel row), is rounded to the 16-bit operand
//       type and parked pixel-major in the wave's private LDS tile; the NEXT group's loads are issued before the math of this one;
//     * per 16-pixel tile, 12 MFMAs (16x16x32) give the three 1x1 convs in the orientation each consumer wants:
//         A, B as D[pixel][channel]  (operands: X^T tile as A, W rows as B) -> a lane holds 4 consecutive pixels of one channel,
//                                    which IS the operand layout of the contraction over pixels (G = A E^T, tile pairs = K of 32);
//         V    as D[channel][pixel]  (operands swapped: same fragments)      -> a lane holds 4 consecutive channels of one pixel:
//                                    the softmax over channels is an in-lane reduction + two xor-shuffles, and the result leaves as
//                                    ONE 8-byte LDS write into the image's pixel-major V tile (HW x 64 B, stays in LDS);
//     * softmax over HW for B is online: running per-channel maximum (log2 units: log2 e is folded into WB / bB, WV / bV), E = 2^(b - max)
//       feeds the G update, the wave's partial G and row sums are rescaled lane-locally when the maximum moves (a G column and
//       its statistics live in the same lane);
//   then the eight partial (G, max, sum) are merged through LDS, M' = WP G is formed (C x 32, fp32 FMAs), and y = M' V + bP is
//   computed from the V tile in LDS (A = V^T rows from LDS, B = M' rows) as D[pixel][channel]: a lane holds 4 consecutive pixels of
//   one output channel = one 16-byte store; a wave's two tiles complete 128-byte lines of y.
// HBM traffic: x once, y once -- nothing else (no workspace).
#include "common.h"
#include "mma.h"

namespace {

constexpr int DS_C = 32;          // c_m = c_n
constexpr int DS_GPX = 32;        // pixels per group (two 16-pixel MFMA tiles = K of the contraction over pixels)

struct DsArgs {
    const float* x; float* y;
    const float* wA; const float* bA; const float* wB; const float* bB; const float* wV; const float* bV; const float* wP; const float* bP;
    int HW;
};

template <int PREC, int C>
__global__ __launch_bounds__(512) void da_small_kernel(const DsArgs a) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int KS = C / 32;                       // k-steps of the 1x1 convs
    constexpr int XP = C * 2 + 16;                   // byte pitch of a pixel row in the wave's X^T tile (16-byte aligned, off the bank period)
    constexpr int NLD = C / 8;                       // 16-byte loads per lane and group (8 channel rows per instruction)
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // layout: [V tile: HW x 64 B] [8 x X^T tiles: 32 x XP] ... the X^T tiles are dead after the streaming phase and are re-used for the
    // merge area (8 x (32 x 32 fp32 G + 32 max + 32 sum)), the merged G (4 KB) and M' (C x 64 B)
    unsigned char* s_v = lds;
    unsigned char* s_x = lds + (size_t)a.HW * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, HW = a.HW;
    const float* xb = a.x + (long)b * C * HW;
    unsigned char* xt = s_x + w * (DS_GPX * XP);

    // ---- weight fragments, kept in registers for the whole image.  B-operand form (column = out channel l15, k = 8 consecutive input
    //      channels) for WA / WB; the same lane -> (row, k) map as the A operand for WV.  log2 e folded into WB, WV and their biases.
    auto wfrag = [&](const float* wrow, float s) {
        const f4 lo = *reinterpret_cast<const f4*>(wrow) * s, hi = *reinterpret_cast<const f4*>(wrow + 4) * s;
        const v4 h0 = M_::cvt(lo), h1 = M_::cvt(hi);
        return v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    };
    v8 fwA[2][KS], fwB[2][KS], fwV[2][KS];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = ct * 16 + l15, k0 = ks * 32 + g * 8;
            fwA[ct][ks] = wfrag(a.wA + (long)o * C + k0, 1.0f);
            fwB[ct][ks] = wfrag(a.wB + (long)o * C + k0, LOG2E);
            fwV[ct][ks] = wfrag(a.wV + (long)o * C + k0, LOG2E);
        }
    float cbA[2], cbB[2];                            // D[pixel][channel]: the bias of column l15 is the same in all four accumulator rows
    f4 rbV[2];                                       // D[channel][pixel]: rows g*4 + r
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        cbA[ct] = a.bA[ct * 16 + l15];
        cbB[ct] = a.bB[ct * 16 + l15] * LOG2E;
        rbV[ct] = *reinterpret_cast<const f4*>(a.bV + ct * 16 + g * 4) * LOG2E;
    }

    // ---- streaming phase ---------------------------------------------------------------------------------------------------------
    const int ngroups = HW / DS_GPX;
    const int chs = lane >> 3, pq = lane & 7;        // load geometry: instruction i covers channels 8 i + chs, pixels 4 pq .. 4 pq + 3
    f4 xr[NLD];
    auto load_group = [&](int grp) {
        const float* p = xb + (long)chs * HW + grp * DS_GPX + pq * 4;
#pragma unroll
        for (int i = 0; i < NLD; ++i) xr[i] = *reinterpret_cast<const f4*>(p + (long)(8 * i) * HW);
    };
    f4 gacc[2][2];                                   // partial G: [it][jt], lane = G[i = it*16 + g*4 + r][j = jt*16 + l15]
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) gacc[it][jt] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};      // channel jt*16 + l15 of B (identical in the four lane groups)

    int grp = w;
    if (grp < ngroups) load_group(grp);
    for (; grp < ngroups; grp += 8) {
        // park the group pixel-major in 16 bit: lane writes 4 pixels x 1 channel per load
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = 8 * i + chs;
            const v4 h = M_::cvt(xr[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<el*>(xt + (pq * 4 + e) * XP + ch * 2) = h[e];
        }
        if (grp + 8 < ngroups) load_group(grp + 8);                       // next group's rows fly under this group's math
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f4 accA[2][2], accB[2][2];                                       // [pixel tile][channel tile]
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            v8 xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const v8*>(xt + (pt * 16 + l15) * XP + (ks * 32 + g * 8) * 2);
            f4 accV[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                accA[pt][ct] = f4{cbA[ct], cbA[ct], cbA[ct], cbA[ct]};
                accB[pt][ct] = f4{cbB[ct], cbB[ct], cbB[ct], cbB[ct]};
                accV[ct] = rbV[ct];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    accA[pt][ct] = M_::mma(xf[ks], fwA[ct][ks], accA[pt][ct]);
                    accB[pt][ct] = M_::mma(xf[ks], fwB[ct][ks], accB[pt][ct]);
                    accV[ct] = M_::mma(fwV[ct][ks], xf[ks], accV[ct]);
                }
            }
            // V: softmax over the 32 channels of pixel l15 (8 values in this lane, the rest in the other three lane groups)
            float vm = fmaxf(fmaxf(fmaxf(accV[0].x, accV[0].y), fmaxf(accV[0].z, accV[0].w)), fmaxf(fmaxf(accV[1].x, accV[1].y), fmaxf(accV[1].z, accV[1].w)));
            vm = fmaxf(vm, __shfl_xor(vm, 16, WAVE));
            vm = fmaxf(vm, __shfl_xor(vm, 32, WAVE));
            f4 e0, e1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e0[r] = __builtin_amdgcn_exp2f(accV[0][r] - vm);
                e1[r] = __builtin_amdgcn_exp2f(accV[1][r] - vm);
            }
            float vs = ((e0.x + e0.y) + (e0.z + e0.w)) + ((e1.x + e1.y) + (e1.z + e1.w));
            vs += __shfl_xor(vs, 16, WAVE);
            vs += __shfl_xor(vs, 32, WAVE);
            const float vinv = 1.0f / vs;
            unsigned char* vrow = s_v + (size_t)(grp * DS_GPX + pt * 16 + l15) * 64;
            *reinterpret_cast<v4*>(vrow + (g * 4) * 2) = M_::cvt(e0 * vinv);
            *reinterpret_cast<v4*>(vrow + (16 + g * 4) * 2) = M_::cvt(e1 * vinv);
        }
        // B: online softmax over pixels.  Group maximum of channel jt*16 + l15: 8 values in this lane, the rest in the other lane groups
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const f4 b0 = accB[0][jt], b1 = accB[1][jt];
            float mx = fmaxf(fmaxf(fmaxf(b0.x, b0.y), fmaxf(b0.z, b0.w)), fmaxf(fmaxf(b1.x, b1.y), fmaxf(b1.z, b1.w)));
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            const float mn = fmaxf(m_run[jt], mx);
            const float sc = __builtin_amdgcn_exp2f(m_run[jt] - mn);        // 2^(-inf) = 0 on the first group: nothing accumulated yet
            m_run[jt] = mn;
            l_run[jt] *= sc;
            gacc[0][jt] = gacc[0][jt] * sc;
            gacc[1][jt] = gacc[1][jt] * sc;
            f4 p0, p1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0[r] = __builtin_amdgcn_exp2f(b0[r] - mn);
                p1[r] = __builtin_amdgcn_exp2f(b1[r] - mn);
            }
            l_run[jt] += ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w));
            accB[0][jt] = p0;
            accB[1][jt] = p1;
        }
        // G[i][j] += sum over the group's 32 pixels of A[i][p] E[j][p]: k enumerates (pixel tile, lane group, r) in both operands
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const v4 a0 = M_::cvt(accA[0][it]), a1 = M_::cvt(accA[1][it]);
            const v8 af = v8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const v4 q0 = M_::cvt(accB[0][jt]), q1 = M_::cvt(accB[1][jt]);
                const v8 ef = v8{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                gacc[it][jt] = M_::mma(af, ef, gacc[it][jt]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the pixels a lane group saw: sum the row sums over the four lane groups (the maximum is already common)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        l_run[jt] += __shfl_xor(l_run[jt], 16, WAVE);
        l_run[jt] += __shfl_xor(l_run[jt], 32, WAVE);
    }
    __syncthreads();                                                         // every wave is done with its X^T tile: the area changes hands
    float* s_gw = reinterpret_cast<float*>(s_x);                             // [8][32*32] partial G
    float* s_mw = s_gw + 8 * 1024;                                           // [8][32] maxima
    float* s_lw = s_mw + 8 * 32;                                             // [8][32] sums
    float* s_g = s_lw + 8 * 32;                                              // [32][32] merged, normalised G
    unsigned char* s_m = reinterpret_cast<unsigned char*>(s_g + 1024);       // M' [C][32] in the operand type
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_gw[w * 1024 + (it * 16 + g * 4 + r) * 32 + jt * 16 + l15] = gacc[it][jt][r];
    if (g == 0) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            s_mw[w * 32 + jt * 16 + l15] = m_run[jt];
            s_lw[w * 32 + jt * 16 + l15] = l_run[jt];
        }
    }
    __syncthreads();
    // merge the eight partials (fixed order: the result does not depend on timing); a wave that saw no group carries max = -inf, sum = 0
    for (int q = t; q < 1024; q += 512) {
        const int j = q & 31;
        float M = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) M = fmaxf(M, s_mw[ww * 32 + j]);
        float L = 0.f, G = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            const float sc = __builtin_amdgcn_exp2f(s_mw[ww * 32 + j] - M);
            L += s_lw[ww * 32 + j] * sc;
            G += s_gw[ww * 1024 + q] * sc;
        }
        s_g[q] = G / L;
    }
    __syncthreads();
    // M'[c][j] = sum_i WP[c][i] G[i][j]   (C x 32 outputs, 32 fp32 FMAs each)
    for (int q = t; q < C * 32; q += 512) {
        const int c = q >> 5, j = q & 31;
        const float* wp = a.wP + (long)c * DS_C;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) acc = __builtin_fmaf(wp[i], s_g[i * 32 + j], acc);
        *reinterpret_cast<el*>(s_m + q * 2) = M_::cvt1(acc);
    }
    __syncthreads();
    // ---- y = M' V + bP from the V tile in LDS, D[pixel][channel]: lane = 4 consecutive pixels of output channel ct*16 + l15 ----------
    v8 mf[C / 16];
    float bp[C / 16];
#pragma unroll
    for (int ct = 0; ct < C / 16; ++ct) {
        mf[ct] = *reinterpret_cast<const v8*>(s_m + (size_t)(ct * 16 + l15) * 64 + g * 16);
        bp[ct] = a.bP[ct * 16 + l15];
    }
    float* yb = a.y + (long)b * C * HW;
    for (int gg = w; gg < ngroups; gg += 8) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int px0 = gg * DS_GPX + pt * 16;
            const v8 vf = *reinterpret_cast<const v8*>(s_v + (size_t)(px0 + l15) * 64 + g * 16);
#pragma unroll
            for (int ct = 0; ct < C / 16; ++ct) {
                const f4 o = M_::mma(vf, mf[ct], f4{bp[ct], bp[ct], bp[ct], bp[ct]});
                __builtin_nontemporal_store(o, reinterpret_cast<f4*>(yb + (long)(ct * 16 + l15) * HW + px0 + g * 4));
            }
        }
    }
}

}  // namespace

namespace mi355 {

bool double_attn_small_ok(int B, int C, int cm, int cn, int HW, int precision) {
    return (precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16) && cm == DS_C && cn == DS_C && C == 64 &&
           HW >= DS_GPX && (HW % DS_GPX) == 0 && HW <= 1024 && B > 0;
}

int double_attn_small(const float* x, const float* wA, const float* bA, const float* wB, const float* bB, const float* wV, const float* bV,
                      const float* wP, const float* bP, float* y, int B, int C, int HW, int precision, hipStream_t st) {
    DsArgs a{x, y, wA, bA, wB, bB, wV, bV, wP, bP, HW};
    // V tile + max(the eight X^T tiles, the merge area)
    const size_t xt = (size_t)8 * DS_GPX * (C * 2 + 16);
    const size_t merge = (size_t)(8 * 1024 + 8 * 32 * 2 + 1024) * 4 + (size_t)C * 64;
    const size_t smem = (size_t)HW * 64 + (xt > merge ? xt : merge);
#define DS_GO(P_, C_)                                                                                       \
    do {                                                                                                    \
        if (int rc = func_dynamic_lds(reinterpret_cast<const void*>(da_small_kernel<P_, C_>), (int)smem)) return rc; \
        da_small_kernel<P_, C_><<<B, 512, smem, st>>>(a);                                                   \
    } while (0)
    if (precision == MI355_PREC_FP16) DS_GO(1, 64);
    else                              DS_GO(2, 64);
#undef DS_GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MI355_EHIP, "mi355_double_attn_fwd: kernel launch -> %s", hipGetErrorString(e));
    return MI355_OK;
}

}  // namespace mi355
