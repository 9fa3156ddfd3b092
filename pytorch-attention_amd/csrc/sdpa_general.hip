// sdpa_general.hip -- softmax(Q K^T * scale + bias) V for any (N_q, N_kv) on gfx950: online softmax over 64-key tiles.
//
// Serves the plain multi-head attention pattern the reference repeats in a dozen files: setr.py:62-72, pvt.py:73-91
// and segformer.py:33-50 (keys / values from a spatially reduced token grid, N_kv << N_q), cmt.py:93-111 (+ an additive
// relative-position term), moat.py:74-84, bvit.py:66-76 ...  The short-sequence kernel of attn.hip keeps every key of a head in
// LDS at once (N <= 224); this one streams K and V:
//
//   workgroup = 4 waves = 64 queries of one (image, head); each wave owns one 16-query tile, Q fragments stay in registers
//   per 64-key tile:  K -> LDS [key][d], V -> LDS transposed [d][key]   (global loads of tile j+1 are in flight while tile j
//                     is computed: registers -> LDS after the compute, two barriers per tile)
//                     S^T = K . Q^T (MFMA 16x16x32; a lane holds 4 consecutive keys of one query per 16-key tile)
//                     running max / sum per query (in-lane over 16 values + two xor-shuffles across the 4 lane groups),
//                     P re-packed in-lane as the A operand, O = O * alpha + P . V
//   end:              O / sum -> per-wave LDS slab -> row-contiguous stores
// q, k, v are addressed through row strides, so they can be slices of one fused projection or separate tensors.
// Head widths above 64 (128 / 192 / 256: ViT.py:68-77 defaults to dim 768 / 4 heads = 192) run with the logits contracted over the
// full width DQK and the value / output columns cut into DV = 64 wide slices, one workgroup per (query block, slice): a slice
// recomputes the logits of its queries (the S^T product is 1/(1 + DQK/DV) ... of the work) but keeps O at 16 registers per lane, and
// the slices of one query block sit next to each other in the grid, i.e. on one XCD, and share K through its L2.
#include "common.h"
#include "mma.h"
#include <type_traits>

namespace {

struct SdpaArgs {
    const void* q; const void* k; const void* v; void* out;
    const float* bias;               // (heads, Nq, Nkv) fp32 or null; image b uses bias + b * bias_bstride
    long bias_bstride;
    int Nq, Nkv, heads;
    int hd, nsl;                     // full head width (column stride between heads); value / output slices per head (hd / DV)
    long ldq, ldk, ldv, ldo;         // row strides in elements
    float scale;
};

constexpr int KTILE = 64;            // keys per streamed tile
constexpr int NWV = 4;               // waves per workgroup

template <int PREC, int D, int DV, bool IO16>
__global__ __launch_bounds__(NWV * 64) void sdpa_stream_kernel(const SdpaArgs a) {
    constexpr int NTHR = NWV * 64;
    static_assert(!IO16 || PREC != 0, "16-bit I/O exists for the fp16 / bf16 operand modes only");
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int NS = M_::NSPLIT;
    constexpr int KP = D + 8;                 // K row pitch (elements)
    constexpr int VP = KTILE + 4;             // V^T row pitch
    constexpr int OP = IO16 ? DV + 8 : DV + 4;  // output slab pitch
    constexpr int K_EL = KTILE * KP, V_EL = DV * VP;
    using slab_t = typename std::conditional<IO16, unsigned short, float>::type;
    using gel = typename std::conditional<IO16, el, float>::type;
    __shared__ __attribute__((aligned(16))) unsigned short s_k[NS * K_EL];
    __shared__ __attribute__((aligned(16))) unsigned short s_v[NS * V_EL];
    __shared__ __attribute__((aligned(16))) slab_t s_o[NWV * 16 * OP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int nqb = (a.Nq + 63) >> 6;
    int bid = xcd_contiguous_block();                       // query blocks of one head (same K / V) and neighbouring heads share an L2
    int sl = 0;
    if constexpr (DV != D) { sl = bid % a.nsl; bid /= a.nsl; }
    const int qb = bid % nqb; bid /= nqb;
    const int head = bid % a.heads;
    const int b = bid / a.heads;
    const gel* qbase = static_cast<const gel*>(a.q) + (long)b * a.Nq * a.ldq + head * a.hd;
    const gel* kbase = static_cast<const gel*>(a.k) + (long)b * a.Nkv * a.ldk + head * a.hd;
    const gel* vbase = static_cast<const gel*>(a.v) + (long)b * a.Nkv * a.ldv + head * a.hd + sl * DV;
    const float* bias = a.bias ? a.bias + (long)b * a.bias_bstride + (long)head * a.Nq * a.Nkv : nullptr;
    const float L2E = 1.44269504088896340736f;

    // ---- Q fragments of this wave's 16 queries (B operand of S^T = K . Q^T): column q = l15, k = d = ks*32 + g*8 + [0,8) ----
    const int q0 = qb * 64 + wave * 16;
    const int qs = q0 + l15;
    v8 qf[D / 32][NS];
    {
        const gel* qrow = qbase + (long)(qs < a.Nq ? qs : 0) * a.ldq;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            if constexpr (IO16) {
                qf[ks][0] = (qs < a.Nq) ? *reinterpret_cast<const v8*>(qrow + ks * 32 + g * 8) : v8{};
            } else {
                f4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
                if (qs < a.Nq) {
                    lo4 = *reinterpret_cast<const f4*>(qrow + ks * 32 + g * 8);
                    hi4 = *reinterpret_cast<const f4*>(qrow + ks * 32 + g * 8 + 4);
                }
                const v4 h0 = M_::cvt(lo4), h1 = M_::cvt(hi4);
                qf[ks][0] = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                if constexpr (NS == 2) {
                    const v4 e0 = M_::cvt_lo(lo4, h0), e1 = M_::cvt_lo(hi4, h1);
                    qf[ks][1] = v8{e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                }
            }
        }
    }

    // ---- staging registers of one K / V tile ----------------------------------------------------------------------------------
    constexpr int EPV = IO16 ? 8 : 4;                          // elements per 16-byte global load
    constexpr int DK_ = D / EPV;                               // vector columns per K row
    constexpr int DV_ = DV / EPV;                              // vector columns per V row (of this slice)
    constexpr int NKI = (KTILE * DK_ + NTHR - 1) / NTHR;       // K vectors per thread
    constexpr int NVI = ((KTILE / 4) * DV_ + NTHR - 1) / NTHR; // V (4 keys x one vector column) groups per thread
    using gv = typename std::conditional<IO16, v8, f4>::type;
    gv kreg[NKI], vreg[NVI][4];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int it = 0; it < NKI; ++it) {
            const int idx = t + it * NTHR, key = idx / DK_, dc = idx % DK_;
            kreg[it] = gv{};
            if (idx < KTILE * DK_ && key0 + key < a.Nkv)
                kreg[it] = *reinterpret_cast<const gv*>(kbase + (long)(key0 + key) * a.ldk + dc * EPV);
        }
#pragma unroll
        for (int it = 0; it < NVI; ++it) {
            const int idx = t + it * NTHR, kg = idx / DV_, dc = idx % DV_;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = key0 + kg * 4 + j;
                vreg[it][j] = gv{};
                if (idx < (KTILE / 4) * DV_ && key < a.Nkv) vreg[it][j] = *reinterpret_cast<const gv*>(vbase + (long)key * a.ldv + dc * EPV);
            }
        }
    };
    auto commit = [&]() {                                      // staging registers -> LDS in the MFMA operand format
#pragma unroll
        for (int it = 0; it < NKI; ++it) {
            const int idx = t + it * NTHR, key = idx / DK_, dc = idx % DK_;
            if (idx < KTILE * DK_) {
                if constexpr (IO16) {
                    *reinterpret_cast<v8*>(s_k + key * KP + dc * 8) = kreg[it];
                } else {
                    const v4 h = M_::cvt(kreg[it]);
                    *reinterpret_cast<v4*>(s_k + key * KP + dc * 4) = h;
                    if constexpr (NS == 2) *reinterpret_cast<v4*>(s_k + K_EL + key * KP + dc * 4) = M_::cvt_lo(kreg[it], h);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NVI; ++it) {
            const int idx = t + it * NTHR, kg = idx / DV_, dc = idx % DV_;
            if (idx < (KTILE / 4) * DV_) {
                if constexpr (IO16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        *reinterpret_cast<v4*>(s_v + (dc * 8 + e) * VP + kg * 4) = v4{vreg[it][0][e], vreg[it][1][e], vreg[it][2][e], vreg[it][3][e]};
                } else {
                    const f4 c[4] = {{vreg[it][0].x, vreg[it][1].x, vreg[it][2].x, vreg[it][3].x},
                                     {vreg[it][0].y, vreg[it][1].y, vreg[it][2].y, vreg[it][3].y},
                                     {vreg[it][0].z, vreg[it][1].z, vreg[it][2].z, vreg[it][3].z},
                                     {vreg[it][0].w, vreg[it][1].w, vreg[it][2].w, vreg[it][3].w}};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v4 h = M_::cvt(c[e]);
                        *reinterpret_cast<v4*>(s_v + (dc * 4 + e) * VP + kg * 4) = h;
                        if constexpr (NS == 2) *reinterpret_cast<v4*>(s_v + V_EL + (dc * 4 + e) * VP + kg * 4) = M_::cvt_lo(c[e], h);
                    }
                }
            }
        }
    };

    // ---- running state: query column l15 of this lane (max, sum); O rows g*4 + r ------------------------------------------------
    float m_run = -INFINITY, l_run = 0.f;
    f4 o[DV / 16];
#pragma unroll
    for (int nt = 0; nt < DV / 16; ++nt) o[nt] = f4{0.f, 0.f, 0.f, 0.f};
    const float sc = a.scale * L2E;                            // logits in log2 units: exp(x) = 2^(x*log2 e)
    const bool bias_vec = bias && (a.Nkv % 4 == 0) && ((reinterpret_cast<uintptr_t>(bias) & 15u) == 0);

    const int ntiles = (a.Nkv + KTILE - 1) / KTILE;
    fetch(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int key0 = tile * KTILE;
        __syncthreads();                                       // everybody is done reading the previous tile
        commit();
        __syncthreads();
        if (tile + 1 < ntiles) fetch(key0 + KTILE);            // in flight during the MFMAs below

        // S^T tiles: lane holds S^T[key = key0 + kt*16 + g*4 + r][q = l15]
        f4 s[KTILE / 16];
#pragma unroll
        for (int kt = 0; kt < KTILE / 16; ++kt) {
            s[kt] = f4{0.f, 0.f, 0.f, 0.f};
            if (key0 + kt * 16 < a.Nkv) {
#pragma unroll
                for (int ks = 0; ks < D / 32; ++ks) {
                    v8 kf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        kf[sp] = *reinterpret_cast<const v8*>(s_k + sp * K_EL + (kt * 16 + l15) * KP + ks * 32 + g * 8);
                    s[kt] = mma_step<PREC>(kf, qf[ks], s[kt]);
                }
            }
        }
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KTILE / 16; ++kt) {
            const int kb = key0 + kt * 16 + g * 4;
            f4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && qs < a.Nq && kb < a.Nkv) {
                const float* br = bias + (long)qs * a.Nkv + kb;
                if (bias_vec) bv = *reinterpret_cast<const f4*>(br);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = (kb + r < a.Nkv) ? br[r] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = (kb + r < a.Nkv) ? s[kt][r] * sc + bv[r] * L2E : -INFINITY;
                s[kt][r] = x;
                mt = fmaxf(mt, x);
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, WAVE));
        mt = fmaxf(mt, __shfl_xor(mt, 32, WAVE));
        const float m_new = fmaxf(m_run, mt);                  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTILE / 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kt][r] - m_new);
                s[kt][r] = p;
                ps += p;
            }
        ps += __shfl_xor(ps, 16, WAVE);
        ps += __shfl_xor(ps, 32, WAVE);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // rescale O: o[nt][r] belongs to query row g*4 + r, whose alpha lives in lanes with l15 == g*4 + r
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, g * 4 + r, WAVE);
#pragma unroll
        for (int nt = 0; nt < DV / 16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[nt][r] *= ar[r];
        // O += P . V : A = P (row q = l15, k enumerates keys as (tile 2kb, g, r) then (tile 2kb+1, g, r)); B = V^T same enumeration
#pragma unroll
        for (int kb = 0; kb < KTILE / 32; ++kb) {
            if (key0 + kb * 32 < a.Nkv) {
                v8 pf[NS];
                const f4 p0 = s[2 * kb], p1 = s[2 * kb + 1];
                const v4 h0 = M_::cvt(p0), h1 = M_::cvt(p1);
                pf[0] = v8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                if constexpr (NS == 2) {
                    const v4 e0 = M_::cvt_lo(p0, h0), e1 = M_::cvt_lo(p1, h1);
                    pf[1] = v8{e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                }
#pragma unroll
                for (int nt = 0; nt < DV / 16; ++nt) {
                    v8 vf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) {
                        const unsigned short* vr = s_v + sp * V_EL + (nt * 16 + l15) * VP + kb * 32 + g * 4;
                        const v4 a0 = *reinterpret_cast<const v4*>(vr);
                        const v4 a1 = *reinterpret_cast<const v4*>(vr + 16);
                        vf[sp] = v8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    }
                    o[nt] = mma_step<PREC>(pf, vf, o[nt]);
                }
            }
        }
    }

    // ---- normalise, stage through the wave's slab, row-contiguous stores ----------------------------------------------------------
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) inv[r] = 1.0f / __shfl(l_run, g * 4 + r, WAVE);
    slab_t* slab = s_o + wave * 16 * OP;
#pragma unroll
    for (int nt = 0; nt < DV / 16; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float val = o[nt][r] * inv[r];
            if constexpr (IO16) *reinterpret_cast<el*>(slab + (g * 4 + r) * OP + nt * 16 + l15) = M_::cvt1(val);
            else slab[(g * 4 + r) * OP + nt * 16 + l15] = val;
        }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    gel* obase = static_cast<gel*>(a.out) + (long)b * a.Nq * a.ldo + head * a.hd + sl * DV;
    constexpr int LPR = DV / EPV, RPI = 64 / LPR;
#pragma unroll
    for (int it = 0; it < 16 / RPI; ++it) {
        const int r = it * RPI + lane / LPR, cv = (lane % LPR) * EPV;
        if (q0 + r < a.Nq)
            *reinterpret_cast<gv*>(obase + (long)(q0 + r) * a.ldo + cv) = *reinterpret_cast<const gv*>(slab + r * OP + cv);
    }
}

template <int D, int DV, bool IO16>
int launch(SdpaArgs a, int B, int precision, hipStream_t st) {
    a.hd = D; a.nsl = D / DV;
    const long blocks = (long)B * a.heads * ((a.Nq + 63) / 64) * a.nsl;
    if (blocks > 0x7FFFFFFFL) return mi355::fail(MI355_EUNSUPPORTED, "mi355_sdpa_general_fwd: grid too large");
    const int grid = (int)blocks;
    if (precision == 1) sdpa_stream_kernel<1, D, DV, IO16><<<grid, NWV * 64, 0, st>>>(a);
    else if (precision == 2) sdpa_stream_kernel<2, D, DV, IO16><<<grid, NWV * 64, 0, st>>>(a);
    else {
        if constexpr (IO16) return mi355::fail(MI355_EINVAL, "mi355_sdpa_general_fwd: 16-bit I/O needs precision 1 or 2");
        else sdpa_stream_kernel<0, D, DV, false><<<grid, NWV * 64, 0, st>>>(a);
    }
    return MI355_OK;
}

}  // namespace

extern "C" int mi355_sdpa_general_fwd(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int num_heads,
                                      int Nq, int Nkv, int head_dim, long ldq, long ldk, long ldv, long ldo, long bias_batch_stride,
                                      float scale, int io16, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(q && k && v && out && B > 0 && num_heads > 0 && Nq > 0 && Nkv > 0);
    MI355_CHECK_ARG(precision >= 0 && precision <= 2 && (io16 == 0 || io16 == 1));
    const long row = (long)num_heads * head_dim;
    MI355_CHECK_ARG(ldq >= row && ldk >= row && ldv >= row && ldo >= row && bias_batch_stride >= 0);
    const int epv = io16 ? 8 : 4;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || (ldq % epv) || (ldk % epv) || (ldv % epv) || (ldo % epv))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_sdpa_general_fwd: q/k/v/out must be 16-byte aligned with 16-byte aligned rows");
    SdpaArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.bias = bias; a.bias_bstride = bias_batch_stride;
    a.Nq = Nq; a.Nkv = Nkv; a.heads = num_heads; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    if (head_dim == 32) rc = io16 ? launch<32, 32, true>(a, B, precision, st) : launch<32, 32, false>(a, B, precision, st);
    else if (head_dim == 64) rc = io16 ? launch<64, 64, true>(a, B, precision, st) : launch<64, 64, false>(a, B, precision, st);
    else if (head_dim == 128) rc = io16 ? launch<128, 64, true>(a, B, precision, st) : launch<128, 64, false>(a, B, precision, st);
    else if (head_dim == 192) rc = io16 ? launch<192, 64, true>(a, B, precision, st) : launch<192, 64, false>(a, B, precision, st);
    else if (head_dim == 256) rc = io16 ? launch<256, 64, true>(a, B, precision, st) : launch<256, 64, false>(a, B, precision, st);
    else return mi355::fail(MI355_EUNSUPPORTED, "mi355_sdpa_general_fwd: head_dim %d not in {32, 64, 128, 192, 256}", head_dim);
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}
