// layernorm.hip -- nn.LayerNorm over the last axis (eps inside the sqrt, biased variance), fp32 math.
// HBM-bound: a row lives in registers between the statistics and the normalise pass, so each element is read once and
// written once.  LPR lanes cooperate on one row (16 / 32 / 64, so that narrow rows -- C = 64 in CSWin stage 1 -- still
// fill the wave with 4 / 2 rows), NV float4 per lane.  Mean and variance are two-pass (sum, then sum of squared
// deviations), the formulation of the reference's nn.LayerNorm (ViT.py:111-114, cswin.py:139,174, xcit.py:271-283,
// mlp_mixer.py:39-41).  The output is fp32, or directly the 16-bit MFMA operand format of the GEMM that consumes it.
#include "common.h"
#include "mma.h"
#include <type_traits>

namespace {
using v4f = float __attribute__((ext_vector_type(4)));

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

template <typename OT>
__device__ __forceinline__ void store4(OT* p, v4f v) {
    if constexpr (std::is_same<OT, float>::value) {
        *reinterpret_cast<v4f*>(p) = v;
    } else {
        typedef OT o4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<o4*>(p) = o4{(OT)v.x, (OT)v.y, (OT)v.z, (OT)v.w};
    }
}

template <int LPR, int NV, typename OT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, OT* __restrict__ y, long rows, int cols,
                                                       float eps, unsigned* ovf) {
    float rgmax = 0.f;                                  // fp16 range guard (common.h): a large LayerNorm gain can saturate the operand
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int n4 = cols >> 2;
    const float inv = 1.0f / (float)cols;
    for (long r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
        const long row = r0 + rsel;
        const bool rok = row < rows;
        const v4f* xr = reinterpret_cast<const v4f*>(x + (rok ? row : 0) * cols);
        v4f v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = sub + LPR * j;
            v[j] = (rok && i < n4) ? xr[i] : v4f{0.f, 0.f, 0.f, 0.f};
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = group_sum<LPR>(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = sub + LPR * j;
            if (i < n4) {
                const v4f d = v[j] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) * inv + eps);
        if (!rok) continue;
        OT* yr = y + row * cols;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = sub + LPR * j;
            if (i < n4) {
                const v4f ww = reinterpret_cast<const v4f*>(w)[i], bb = reinterpret_cast<const v4f*>(b)[i];
                const v4f o = (v[j] - mean) * rstd * ww + bb;
                if constexpr (std::is_same<OT, _Float16>::value) rgmax = rg_absmax4(rgmax, o);
                store4<OT>(yr + 4 * i, o);
            }
        }
    }
    if constexpr (std::is_same<OT, _Float16>::value) rg_report(rgmax, ovf, 2u);
}

// any width / alignment: three sweeps over the (cache-resident) row
template <typename OT>
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, OT* __restrict__ y, long rows,
                                                               int cols, float eps, unsigned* ovf) {
    float rgmax = 0.f;
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long row = wave0; row < rows; row += nwaves) {
        const float* xr = x + row * cols;
        float s = 0.f;
        for (int i = lane; i < cols; i += 64) s += xr[i];
        const float mean = wave_sum(s) / (float)cols;
        float q = 0.f;
        for (int i = lane; i < cols; i += 64) { const float d = xr[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
        for (int i = lane; i < cols; i += 64) {
            const float o = (xr[i] - mean) * rstd * w[i] + b[i];
            if constexpr (std::is_same<OT, _Float16>::value) rgmax = rg_absmax1(rgmax, o);
            y[row * cols + i] = (OT)o;
        }
    }
    if constexpr (std::is_same<OT, _Float16>::value) rg_report(rgmax, ovf, 2u);
}

template <typename OT>
int launch_ln(const float* x, const float* weight, const float* bias, OT* y, int rows, int cols, float eps, hipStream_t st) {
    unsigned* ovf = std::is_same<OT, _Float16>::value ? mi355::range_word(st) : nullptr;
    MI355_TRACE(st, "layernorm_kernel<%s> rows=%d cols=%d", std::is_same<OT, float>::value ? "out32" : "out16", rows, cols);
    const bool vec = (cols % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(weight) && aligned16(bias);
#define LN(LPR_, NV_)                                                                                                \
    do {                                                                                                             \
        const long waves = ((long)rows + (64 / LPR_) - 1) / (64 / LPR_);                                            \
        const int grid = (int)((waves + 3) / 4 < 8192 ? (waves + 3) / 4 : 8192);                                    \
        layernorm_kernel<LPR_, NV_, OT><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps, ovf);                \
    } while (0)
    if (vec && cols <= 64)        LN(16, 1);
    else if (vec && cols <= 128)  LN(32, 1);
    else if (vec && cols <= 256)  LN(64, 1);
    else if (vec && cols <= 512)  LN(64, 2);
    else if (vec && cols <= 1024) LN(64, 4);
    else if (vec && cols <= 2048) LN(64, 8);
    else {
        const int grid = cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192;
        layernorm_generic_kernel<OT><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps, ovf);
    }
#undef LN
    return MI355_OK;
}
}  // namespace

extern "C" {

int mi355_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, int rows, int cols, float eps,
                        mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && bias && y && rows > 0 && cols > 0);
    launch_ln<float>(x, weight, bias, y, rows, cols, eps, static_cast<hipStream_t>(stream));
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_layernorm16_fwd(const float* x, const float* weight, const float* bias, void* y16, int rows, int cols, float eps,
                          int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && bias && y16 && rows > 0 && cols > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (precision == MI355_PREC_FP16) launch_ln<_Float16>(x, weight, bias, static_cast<_Float16*>(y16), rows, cols, eps, st);
    else                              launch_ln<__bf16>(x, weight, bias, static_cast<__bf16*>(y16), rows, cols, eps, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"

// LayerNorm over channels, written TRANSPOSED per image in the 16-bit operand format: x (B, N, C) fp32 -> ut (B, C, NP), ut[b,c,n] =
// LN(x[b,n,:])[c] for n < N and 0 for N <= n < NP.  This is the K-major activation the Mixer token-mixing product wants as a
// plain row-major GEMM operand (mlp_mixer.py:45-47: norm1 -> transpose(1,2) -> token_mlp).  One workgroup = 32 tokens of one image;
// a wave normalises two tokens at a time (lane c = lane + 64k, coalesced 256-byte loads) and parks the pair as one 32-bit word
// in LDS at [c][pair] with a 17-word pitch (odd: conflict-free for the column writes and the row reads); four lanes then
// write each channel's 64-byte run.
namespace {
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <typename T, int KC>
__global__ __launch_bounds__(256) void layernorm16_t_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           T* __restrict__ ut, int N, int C, int NP, float eps) {
    extern __shared__ unsigned int lds_t[];                      // [C][17]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n0 = blockIdx.x * 32, img = blockIdx.y;
    float wv[KC], bv[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = lane + 64 * k;
        wv[k] = c < C ? w[c] : 0.f; bv[k] = c < C ? b[c] : 0.f;
    }
    const float inv = 1.f / (float)C;
#pragma unroll 1
    for (int pr = 0; pr < 4; ++pr) {
        const int tp = wave * 4 + pr;                            // pair index inside the tile: tokens n0 + 2*tp, +1
        const int na = n0 + 2 * tp, nb = na + 1;
        float va[KC], vb[KC];
        const float* ra = x + ((long)img * N + (na < N ? na : 0)) * C;
        const float* rb = x + ((long)img * N + (nb < N ? nb : 0)) * C;
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = lane + 64 * k;
            va[k] = c < C ? ra[c] : 0.f; vb[k] = c < C ? rb[c] : 0.f;
            sa += va[k]; sb += vb[k];
        }
        sa = wave_sum(sa) * inv; sb = wave_sum(sb) * inv;
        float qa = 0.f, qb = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = lane + 64 * k;
            const float da = c < C ? va[k] - sa : 0.f, db = c < C ? vb[k] - sb : 0.f;
            qa += da * da; qb += db * db;
        }
        const float ia = rsqrtf(wave_sum(qa) * inv + eps), ib = rsqrtf(wave_sum(qb) * inv + eps);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = lane + 64 * k;
            if (c >= C) continue;
            const T ha = na < N ? (T)((va[k] - sa) * ia * wv[k] + bv[k]) : (T)0.f;
            const T hb = nb < N ? (T)((vb[k] - sb) * ib * wv[k] + bv[k]) : (T)0.f;
            lds_t[c * 17 + tp] = (unsigned int)__builtin_bit_cast(unsigned short, ha) | ((unsigned int)__builtin_bit_cast(unsigned short, hb) << 16);
        }
    }
    __syncthreads();
    for (int idx = t; idx < C * 4; idx += 256) {
        const int c = idx >> 2, q = idx & 3;
        const unsigned int* p = lds_t + c * 17 + q * 4;
        u4 o{p[0], p[1], p[2], p[3]};
        *reinterpret_cast<u4*>(ut + ((long)img * C + c) * NP + n0 + q * 8) = o;
    }
}
}  // namespace

extern "C" int mi355_layernorm16_t_fwd(const float* x, const float* weight, const float* bias, void* ut16, int B, int N, int C, int NP,
                                       float eps, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && bias && ut16 && B > 0 && N > 0 && C > 0 && NP >= N);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((NP & 31) || C > 1024 || !aligned16(ut16))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_layernorm16_t_fwd: needs NP %% 32 == 0 and C <= 1024 (NP=%d C=%d)", NP, C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(NP / 32, B);
    const size_t lds = (size_t)C * 17 * 4;
#define LNT(T_, KC_) layernorm16_t_kernel<T_, KC_><<<grid, 256, lds, st>>>(x, weight, bias, static_cast<T_*>(ut16), N, C, NP, eps)
#define LNT_BY_C(T_) do { if (C <= 256) LNT(T_, 4); else if (C <= 512) LNT(T_, 8); else LNT(T_, 16); } while (0)
    if (precision == MI355_PREC_FP16) LNT_BY_C(_Float16); else LNT_BY_C(__bf16);
#undef LNT_BY_C
#undef LNT
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// mean over the token axis: y[b, c] = (1/N) sum_n x[b*batch_stride + n*C + c]   (CSWin head cswin.py:341, Mixer head mlp_mixer.py:77,
// ViT global_pool="avg" ViT.py:189-190 via an offset base pointer).  One workgroup per (image, 256-channel slab).
namespace {
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long batch_stride,
                                                        int slabs) {
    const int b = blockIdx.x / slabs, c = (blockIdx.x % slabs) * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = x + (long)b * batch_stride + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int n = 0;
    for (; n + 3 < N; n += 4) {
        s0 += p[(long)n * C]; s1 += p[(long)(n + 1) * C]; s2 += p[(long)(n + 2) * C]; s3 += p[(long)(n + 3) * C];
    }
    for (; n < N; ++n) s0 += p[(long)n * C];
    y[(long)b * C + c] = ((s0 + s1) + (s2 + s3)) / (float)N;
}
}  // namespace

extern "C" int mi355_token_mean_fwd(const float* x, float* y, int B, int N, int C, long batch_stride, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && B > 0 && N > 0 && C > 0 && batch_stride >= (long)N * C);
    const int slabs = cdiv(C, 256);
    token_mean_kernel<<<B * slabs, 256, 0, static_cast<hipStream_t>(stream)>>>(x, y, N, C, batch_stride, slabs);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}
