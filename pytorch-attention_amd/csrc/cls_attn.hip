// cls_attn.hip -- XCiT class-attention stage: the one-query attention core and the strided axpby glue (gfx950).
//
// Both are tiny next to the XCA blocks (the class-attention stage touches B*N*C floats a handful of times), so they are plain
// fp32 vector kernels: exact arithmetic, coalesced 16-byte accesses, no MFMA.
#include "common.h"

namespace {

using v4f = float __attribute__((ext_vector_type(4)));

// One wave per (image, head).  Phase 1: lanes own keys (n = lane, lane+64, ...) and walk the d contiguous floats of their key;
// phase 2: lanes own output channels j < d and walk the keys (each step reads one coalesced d-float row of V).
template <int D>
__global__ __launch_bounds__(64) void class_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out, int N, int h,
                                                        long ldq, long ldkv, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];      // N probabilities
    __shared__ __attribute__((aligned(16))) float s_q[D];
    const int b = blockIdx.x / h, i = blockIdx.x - b * h, lane = threadIdx.x;
    if (lane < D) s_q[lane] = q[(long)b * ldq + i * D + lane];
    __syncthreads();
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) {
        const v4f* kr = reinterpret_cast<const v4f*>(k + ((long)b * N + n) * ldkv + i * D);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < D / 4; j += 2) {
            const v4f a = kr[j], c = kr[j + 1];
            const v4f qa = reinterpret_cast<const v4f*>(s_q)[j], qc = reinterpret_cast<const v4f*>(s_q)[j + 1];
            s0 += qa.x * a.x; s0 += qa.y * a.y; s0 += qa.z * a.z; s0 += qa.w * a.w;
            s1 += qc.x * c.x; s1 += qc.y * c.y; s1 += qc.z * c.z; s1 += qc.w * c.w;
        }
        const float s = (s0 + s1) * scale;
        s_p[n] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float e = expf(s_p[n] - mx);
        s_p[n] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane < D) {
        const float* vc = v + (long)b * N * ldkv + i * D + lane;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int n = 0;
        for (; n + 3 < N; n += 4) {
            a0 += s_p[n] * vc[(long)n * ldkv];
            a1 += s_p[n + 1] * vc[(long)(n + 1) * ldkv];
            a2 += s_p[n + 2] * vc[(long)(n + 2) * ldkv];
            a3 += s_p[n + 3] * vc[(long)(n + 3) * ldkv];
        }
        for (; n < N; ++n) a0 += s_p[n] * vc[(long)n * ldkv];
        out[((long)b * h + i) * D + lane] = ((a0 + a1) + (a2 + a3)) / sum;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                    const float* __restrict__ gamma, float* __restrict__ y, long rows, int cols,
                                                    long ldx, long ldu, long ldy, float alpha) {
    constexpr int VW = VEC ? 4 : 1;
    const int per_row = cols / VW;
    const long total = rows * per_row;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / per_row;
        const int c = (int)(e - r * per_row) * VW;
        if constexpr (VEC) {
            v4f a = *reinterpret_cast<const v4f*>(x + r * ldx + c) * alpha;
            if (u) {
                const v4f w = *reinterpret_cast<const v4f*>(u + r * ldu + c);
                a += gamma ? *reinterpret_cast<const v4f*>(gamma + c) * w : w;
            }
            *reinterpret_cast<v4f*>(y + r * ldy + c) = a;
        } else {
            float a = x[r * ldx + c] * alpha;
            if (u) a += (gamma ? gamma[c] : 1.0f) * u[r * ldu + c];
            y[r * ldy + c] = a;
        }
    }
}

// Depth-wise "patch" convolution on a token grid (kernel == stride == sr, groups == C): the spatial reduction in front of K / V in
// PVT / CMT (pvt.py:66-70).  One thread per (output token, 4 channels); taps read channel-contiguous float4's.
__global__ __launch_bounds__(256) void dw_patch_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int B, int H, int W, int C, int sr) {
    const int c4 = C >> 2, OH = H / sr, OW = W / sr;
    const long total = (long)B * OH * OW * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % c4) * 4;
        long r = e / c4;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        v4f acc = bias ? *reinterpret_cast<const v4f*>(bias + c) : v4f{0.f, 0.f, 0.f, 0.f};
        const float* xb = x + ((long)b * H * W) * C + c;
        for (int ky = 0; ky < sr; ++ky)
            for (int kx = 0; kx < sr; ++kx) {
                const v4f v = *reinterpret_cast<const v4f*>(xb + ((long)(oy * sr + ky) * W + ox * sr + kx) * C);
                const int tap = ky * sr + kx, kk = sr * sr;
                acc.x += w[(long)c * kk + tap] * v.x;
                acc.y += w[(long)(c + 1) * kk + tap] * v.y;
                acc.z += w[(long)(c + 2) * kk + tap] * v.z;
                acc.w += w[(long)(c + 3) * kk + tap] * v.w;
            }
        *reinterpret_cast<v4f*>(y + (((long)b * OH + oy) * OW + ox) * C + c) = acc;
    }
}

}  // namespace

extern "C" {

int mi355_dwconv_patch_tokens_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int H, int W, int C, int sr,
                                  mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && y && B > 0 && H > 0 && W > 0 && C > 0 && sr > 0);
    if ((C & 3) || (H % sr) || (W % sr) || !aligned16(x) || !aligned16(y) || (bias && !aligned16(bias)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_dwconv_patch_tokens_fwd: C %% 4 == 0, H and W divisible by sr, 16-byte aligned buffers");
    const long work = (long)B * (H / sr) * (W / sr) * (C / 4);
    long blocks = (work + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    dw_patch_kernel<<<(int)blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(x, weight, bias, y, B, H, W, C, sr);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_class_attn_fwd(const float* q, const float* k, const float* v, float* out, int B, int N, int num_heads, int head_dim,
                         long ldq, long ldkv, float scale, mi355_stream_t stream) {
    MI355_CHECK_ARG(q && k && v && out && B > 0 && N > 0 && num_heads > 0);
    MI355_CHECK_ARG(ldq >= (long)num_heads * head_dim && ldkv >= (long)num_heads * head_dim);
    if (N > 4096) return mi355::fail(MI355_EUNSUPPORTED, "mi355_class_attn_fwd: N = %d > 4096", N);
    if (!aligned16(k) || (ldkv & 3) || (head_dim & 7))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_class_attn_fwd: keys must be 16-byte aligned, ldkv %% 4 == 0, head_dim %% 8 == 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t smem = (size_t)((N + 3) & ~3) * sizeof(float);
    const int grid = B * num_heads;
    switch (head_dim) {
        case 16: class_attn_kernel<16><<<grid, 64, smem, st>>>(q, k, v, out, N, num_heads, ldq, ldkv, scale); break;
        case 32: class_attn_kernel<32><<<grid, 64, smem, st>>>(q, k, v, out, N, num_heads, ldq, ldkv, scale); break;
        case 48: class_attn_kernel<48><<<grid, 64, smem, st>>>(q, k, v, out, N, num_heads, ldq, ldkv, scale); break;
        case 64: class_attn_kernel<64><<<grid, 64, smem, st>>>(q, k, v, out, N, num_heads, ldq, ldkv, scale); break;
        default: return mi355::fail(MI355_EUNSUPPORTED, "mi355_class_attn_fwd: head_dim %d not in {16,32,48,64}", head_dim);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_axpby_fwd(const float* x, const float* u, const float* gamma, float* y, long rows, int cols, long ldx, long ldu, long ldy,
                    float alpha, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= 0 && ldu >= 0 && ldy >= cols);
    MI355_CHECK_ARG((ldx == 0 || ldx >= cols) && (!u || ldu == 0 || ldu >= cols));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (cols % 4 == 0) && aligned16(x) && aligned16(y) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                     (!u || (aligned16(u) && ldu % 4 == 0)) && (!gamma || aligned16(gamma));
    const long work = rows * (cols / (vec ? 4 : 1));
    long blocks = (work + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (vec) axpby_kernel<true><<<(int)blocks, 256, 0, st>>>(x, u, gamma, y, rows, cols, ldx, ldu, ldy, alpha);
    else     axpby_kernel<false><<<(int)blocks, 256, 0, st>>>(x, u, gamma, y, rows, cols, ldx, ldu, ldy, alpha);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
