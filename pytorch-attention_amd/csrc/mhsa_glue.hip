// mhsa_glue.hip -- the pieces around the streaming attention core that the remaining copies of the multi-head pattern need
// (SURVEY 8 f1): the depth-wise convolution in front of CvT's qkv projection (cvt.py:48-52), the unscaled logits and top-k
// mask of KVT's k-NN attention (kvt.py:83-88).
#include "common.h"

namespace {

// Depth-wise ks x ks convolution (stride 1, zero padding (ks-1)/2) with a folded BatchNorm, NCHW in, token-major out:
//   y[b, p, c] = bias[c] + sum_{u,v} w[c, u, v] * x[b, c, i + u - pad, j + v - pad],  p = i * W + j
// A workgroup owns 32 pixels x 32 channels: threads read x with the pixel index fastest (coalesced along W), transpose the
// tile through LDS and write rows of 32 channels.
__global__ __launch_bounds__(256) void dwconv_nchw_tokens_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ y, int C, int H, int W, int ks) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long HW = (long)H * W, p = (long)blockIdx.x * 32 + tx;
    const int c0 = blockIdx.y * 32, pad = (ks - 1) / 2;
    const int i = (int)((p < HW ? p : HW - 1) / W), j = (int)((p < HW ? p : HW - 1) - (long)i * W);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cl = ty + r * 8, c = c0 + cl;
        float acc = 0.f;
        if (c < C && p < HW) {
            const float* xc = x + ((long)b * C + c) * HW;
            const float* wc = w + (long)c * ks * ks;
            acc = bias ? bias[c] : 0.f;
            for (int u = 0; u < ks; ++u) {
                const int ii = i + u - pad;
                if (ii < 0 || ii >= H) continue;
                for (int v = 0; v < ks; ++v) {
                    const int jj = j + v - pad;
                    if (jj >= 0 && jj < W) acc = __builtin_fmaf(wc[u * ks + v], xc[(long)ii * W + jj], acc);
                }
            }
        }
        tile[cl][tx] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long pp = (long)blockIdx.x * 32 + ty + r * 8;
        const int c = c0 + tx;
        if (pp < HW && c < C) y[((long)b * HW + pp) * C + c] = tile[tx][ty + r * 8];
    }
}

// P2T's pooled key/value source (p2t.py:76-83) on the token layout: adaptive average pooling of the (H x W) token grid to (OH x OW)
// with ATen's bin edges (start = floor(i * H / OH), end = ceil((i + 1) * H / OH)), channels fastest.
__global__ __launch_bounds__(256) void adaptive_pool_tokens_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int OH,
                                                                  int OW, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long t = idx / C;
    const int oj = (int)(t % OW), oi = (int)((t / OW) % OH);
    const long b = t / ((long)OW * OH);
    const int i0 = (int)(((long)oi * H) / OH), i1 = (int)((((long)oi + 1) * H + OH - 1) / OH);
    const int j0 = (int)(((long)oj * W) / OW), j1 = (int)((((long)oj + 1) * W + OW - 1) / OW);
    const float* xb = x + b * (long)H * W * C + c;
    float s = 0.f;
    for (int i = i0; i < i1; ++i)
        for (int j = j0; j < j1; ++j) s += xb[((long)i * W + j) * C];
    y[idx] = s / (float)((i1 - i0) * (j1 - j0));
}

// y[b, p, c] = x[b, p, c] + bias[c] + sum_uv w[c, u, v] * x[b, (i+u-1, j+v-1), c]: the `pool + l(pool)` of p2t.py:79 with l a depth-wise
// 3x3 convolution, on a (H x W) token grid; y rows live in a longer token sequence (batch stride y_bstride).
__global__ __launch_bounds__(256) void dwconv3x3_tokens_residual_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                       float* __restrict__ y, int H, int W, int C, long y_bstride, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long t = idx / C;
    const int j = (int)(t % W), i = (int)((t / W) % H);
    const long b = t / ((long)W * H);
    const float* xb = x + b * (long)H * W * C + c;
    float acc = xb[((long)i * W + j) * C] + (bias ? bias[c] : 0.f);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int ii = i + u - 1;
        if (ii < 0 || ii >= H) continue;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int jj = j + v - 1;
            if (jj >= 0 && jj < W) acc = __builtin_fmaf(w[c * 9 + u * 3 + v], xb[((long)ii * W + jj) * C], acc);
        }
    }
    y[b * y_bstride + ((long)i * W + j) * C + c] = acc;
}

// Top-k mask of every logits row, in place: entries among the k largest of their row become 0, all others -1e30 (an additive
// attention bias; finite so that an all-masked key tile of the streaming softmax stays NaN-free).  One wave per row; the k-th
// largest value is found by a 32-step bisection on the order-preserving integer image of the floats (count >= candidate).
template <int PER>
__global__ __launch_bounds__(256) void topk_mask_kernel(float* __restrict__ logits, long rows, int N, int k) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = logits + row * N;
    unsigned key[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int idx = lane + 64 * q;
        if (idx < N) {
            const unsigned u = __float_as_uint(p[idx]);
            key[q] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone: larger float <=> larger key
        } else {
            key[q] = 0u;                                               // padding lane: excluded from the count by its index
        }
    }
    unsigned lo = 0u, hi = 0xFFFFFFFFu;                                // largest T with count(key >= T) >= k
    while (lo < hi) {
        const unsigned mid = lo + (unsigned)(((unsigned long long)hi - lo + 1ull) >> 1);
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) cnt += (lane + 64 * q < N && key[q] >= mid) ? 1 : 0;
        cnt = (int)wave_sum((float)cnt);
        if (cnt >= k) lo = mid; else hi = mid - 1u;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int idx = lane + 64 * q;
        if (idx < N) p[idx] = key[q] >= lo ? 0.f : -1e30f;
    }
}

}  // namespace

extern "C" {

int mi355_dwconv_nchw_tokens_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int C, int H, int W, int ks,
                                 mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && y && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && ks >= 1 && (ks & 1));
    dwconv_nchw_tokens_kernel<<<dim3(cdiv((long)H * W, 32), cdiv(C, 32), B), 256, 0, static_cast<hipStream_t>(stream)>>>(x, weight, bias, y, C, H, W,
                                                                                                                       ks);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_qk_logits_fwd(const float* q, const float* k, float* logits, int B, int heads, int Nq, int Nkv, int head_dim, int ldq, int ldk,
                        int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(q && k && logits && B > 0 && heads > 0 && Nq > 0 && Nkv > 0 && head_dim > 0);
    MI355_CHECK_ARG(ldq >= heads * head_dim && ldk >= heads * head_dim);
    MI355_CHECK_ARG(precision >= MI355_PREC_STRICT && precision <= MI355_PREC_BF16);
    if ((head_dim & 3) || (ldq & 3) || (ldk & 3) || !aligned16(q) || !aligned16(k))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_qk_logits_fwd: head_dim and the row strides must be multiples of 4 (d=%d)", head_dim);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int h = 0; h < heads; ++h) {       // one batched NT product per head: batch = images, the head selects a column block of q / k
        const int rc = mi355::gemm_nt_batched(q + (long)h * head_dim, k + (long)h * head_dim, logits + (long)h * Nq * Nkv, B, Nq, Nkv, head_dim, ldq,
                                              ldk, Nkv, (long)Nq * ldq, (long)Nkv * ldk, (long)heads * Nq * Nkv, precision, st);
        if (rc) return rc;
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_adaptive_pool_tokens_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0);
    const long total = (long)B * OH * OW * C;
    adaptive_pool_tokens_kernel<<<cdiv(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(x, y, H, W, C, OH, OW, total);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_dwconv3x3_tokens_residual_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int H, int W, int C,
                                        long y_batch_stride, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && y && B > 0 && H > 0 && W > 0 && C > 0 && y_batch_stride >= (long)H * W * C);
    const long total = (long)B * H * W * C;
    dwconv3x3_tokens_residual_kernel<<<cdiv(total, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(x, weight, bias, y, H, W, C, y_batch_stride,
                                                                                                     total);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_topk_mask_fwd(float* logits, long rows, int N, int k, mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && rows > 0 && N > 0 && k > 0 && k <= N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = cdiv(rows, 4);
    if (N <= 256)       topk_mask_kernel<4><<<grid, 256, 0, st>>>(logits, rows, N, k);
    else if (N <= 1024) topk_mask_kernel<16><<<grid, 256, 0, st>>>(logits, rows, N, k);
    else if (N <= 4096) topk_mask_kernel<64><<<grid, 256, 0, st>>>(logits, rows, N, k);
    else return mi355::fail(MI355_EUNSUPPORTED, "mi355_topk_mask_fwd: row length %d > 4096", N);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
