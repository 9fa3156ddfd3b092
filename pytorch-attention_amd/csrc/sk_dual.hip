// sk_dual.hip -- the last three members of the reference's attention zoo: SKLayer (sk_module.py:17-56) and the two
// halves of DANet's dual attention, PAM / CAM (dual_attention.py:10-42).
//
// SKLayer: both grouped 3x3 branches (dilation 1 and 2) read the same input, so one kernel computes them together, applies the
//   folded BatchNorm + ReLU, writes u1 / u2 and leaves the pooled sum of u1 + u2 per (image, channel) -- a workgroup owns one
//   (image, group) and therefore every pixel of its output channels, which makes the pooled sum a plain deterministic block
//   reduction.  A tiny per-image kernel evaluates fc -> BN1d -> ReLU -> fc1 / fc2 -> softmax over the two branches, and the
//   apply pass blends u1 and u2.
// CAM: softmax(X X^T) X on the GEMM engine (x is a (C x HW) matrix per image, HW contiguous -- never transposed in memory), the
//   row softmax scaled by beta in between, the residual in the second product's epilogue.
// PAM: host side composes the three 1x1 convs (one token-major GEMM), the streaming attention kernel and tokens_to_nchw_axpy_kernel
//   below (y[b,c,p] = alpha * tok[b,p,c] + x[b,c,p], transposed through LDS).
#include "common.h"
#include "mma.h"

namespace {

// ---- SK: two grouped 3x3 convolutions + BN + ReLU + pooled sum -----------------------------------------------------------------
// grid (groups, B), 256 threads; thread owns PIX consecutive pixels and the COG output channels of its group for both branches.
// w1 / w2: (planes, cin_g, 3, 3); sc / sh: folded BatchNorm (+ conv bias) per output channel.
template <int COG, int PIX>
__global__ __launch_bounds__(256) void sk_branch_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ sc1,
                                                       const float* __restrict__ sh1, const float* __restrict__ w2, const float* __restrict__ sc2,
                                                       const float* __restrict__ sh2, float* __restrict__ u1, float* __restrict__ u2,
                                                       float* __restrict__ pooled, int Cin, int planes, int cin_g, int H, int W) {
    __shared__ float red[4][COG];
    const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long HW = (long)H * W;
    const float* xg = x + ((long)b * Cin + (long)g * cin_g) * HW;
    const int co0 = g * COG;
    const float* wa = w1 + (long)co0 * cin_g * 9;
    const float* wb = w2 + (long)co0 * cin_g * 9;
    float psum[COG];
#pragma unroll
    for (int o = 0; o < COG; ++o) psum[o] = 0.f;
    for (long p0 = (long)t * PIX; p0 < HW; p0 += 256 * PIX) {
        int pi[PIX], pj[PIX];
#pragma unroll
        for (int e = 0; e < PIX; ++e) {
            const long p = p0 + e < HW ? p0 + e : HW - 1;
            pi[e] = (int)(p / W); pj[e] = (int)(p - (long)pi[e] * W);
        }
        f4 a1[COG], a2[COG];                                     // PIX <= 4 lanes of the vector are used
#pragma unroll
        for (int o = 0; o < COG; ++o) { a1[o] = f4{0.f, 0.f, 0.f, 0.f}; a2[o] = f4{0.f, 0.f, 0.f, 0.f}; }
        for (int ci = 0; ci < cin_g; ++ci) {
            const float* xc = xg + (long)ci * HW;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    f4 x1{0.f, 0.f, 0.f, 0.f}, x2{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < PIX; ++e) {
                        const int i1 = pi[e] + (u - 1), j1 = pj[e] + (v - 1);
                        const int i2 = pi[e] + 2 * (u - 1), j2 = pj[e] + 2 * (v - 1);
                        x1[e] = (i1 >= 0 && i1 < H && j1 >= 0 && j1 < W) ? xc[(long)i1 * W + j1] : 0.f;
                        x2[e] = (i2 >= 0 && i2 < H && j2 >= 0 && j2 < W) ? xc[(long)i2 * W + j2] : 0.f;
                    }
#pragma unroll
                    for (int o = 0; o < COG; ++o) {
                        const long wi = ((long)o * cin_g + ci) * 9 + u * 3 + v;
                        a1[o] += wa[wi] * x1;
                        a2[o] += wb[wi] * x2;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < COG; ++o) {
            const float s1 = sc1[co0 + o], t1 = sh1[co0 + o], s2 = sc2[co0 + o], t2 = sh2[co0 + o];
            float* o1 = u1 + ((long)b * planes + co0 + o) * HW + p0;
            float* o2 = u2 + ((long)b * planes + co0 + o) * HW + p0;
#pragma unroll
            for (int e = 0; e < PIX; ++e) {
                if (p0 + e < HW) {
                    const float r1 = fmaxf(a1[o][e] * s1 + t1, 0.f), r2 = fmaxf(a2[o][e] * s2 + t2, 0.f);
                    o1[e] = r1; o2[e] = r2;
                    psum[o] += r1 + r2;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COG; ++o) {
        const float s = wave_sum(psum[o]);
        if (lane == 0) red[wave][o] = s;
    }
    __syncthreads();
    if (t < COG) pooled[(long)b * planes + co0 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// fc -> BatchNorm1d (folded) -> ReLU -> fc1 / fc2 -> softmax over the two branches; one workgroup per image.  att: (B, 2, planes)
__global__ __launch_bounds__(256) void sk_select_kernel(const float* __restrict__ pooled, const float* __restrict__ wf, const float* __restrict__ bf,
                                                       const float* __restrict__ bn_s, const float* __restrict__ bn_t, const float* __restrict__ wa,
                                                       const float* __restrict__ ba, const float* __restrict__ wb, const float* __restrict__ bb,
                                                       float* __restrict__ att, int planes, int d, float inv_hw) {
    extern __shared__ float z[];                                 // d
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* s = pooled + (long)b * planes;
    for (int k = wave; k < d; k += 4) {
        float acc = 0.f;
        for (int c = lane; c < planes; c += 64) acc = __builtin_fmaf(wf[(long)k * planes + c], s[c] * inv_hw, acc);
        acc = wave_sum(acc);
        if (lane == 0) z[k] = fmaxf((acc + bf[k]) * bn_s[k] + bn_t[k], 0.f);
    }
    __syncthreads();
    for (int c = t; c < planes; c += 256) {
        float a = ba[c], bq = bb[c];
        for (int k = 0; k < d; ++k) {
            a = __builtin_fmaf(wa[(long)c * d + k], z[k], a);
            bq = __builtin_fmaf(wb[(long)c * d + k], z[k], bq);
        }
        const float m = fmaxf(a, bq), ea = expf(a - m), eb = expf(bq - m), inv = 1.0f / (ea + eb);
        att[((long)b * 2 + 0) * planes + c] = ea * inv;
        att[((long)b * 2 + 1) * planes + c] = eb * inv;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void sk_apply_kernel(const float* __restrict__ u1, const float* __restrict__ u2, const float* __restrict__ att,
                                                      float* __restrict__ y, long total, long HW, int planes) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long per = HW / VEC, plane = idx / per, off = plane * HW + (idx - plane * per) * VEC;
    const long img = plane / planes, c = plane - img * planes;
    const float a = att[(img * 2 + 0) * planes + c], b = att[(img * 2 + 1) * planes + c];
    if constexpr (VEC == 4) {
        const f4 p = *reinterpret_cast<const f4*>(u1 + off), q = *reinterpret_cast<const f4*>(u2 + off);
        *reinterpret_cast<f4*>(y + off) = p * a + q * b;
    } else {
        y[off] = u1[off] * a + u2[off] * b;
    }
}

// ---- CAM: in-place row softmax of the (C x C) Gram matrices, scaled by *scale --------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_scaled_kernel(float* __restrict__ p, int cols, long total_rows, const float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total_rows) return;
    float* row = p + r * cols;
    float m = -INFINITY;
    for (int i = lane; i < cols; i += 64) m = fmaxf(m, row[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += expf(row[i] - m);
    s = wave_sum(s);
    const float k = scale[0] / s;
    for (int i = lane; i < cols; i += 64) row[i] = expf(row[i] - m) * k;
}

// ---- PAM epilogue: y[b, c, p] = alpha[0] * tok[b, p, c] + x[b, c, p]; 32 x 32 tiles through LDS ----------------------------------
__global__ __launch_bounds__(256) void tokens_to_nchw_axpy_kernel(const float* __restrict__ tok, const float* __restrict__ x, const float* __restrict__ alpha,
                                                                 float* __restrict__ y, long HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const float* tb = tok + (long)b * HW * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long p = p0 + ty + r * 8;
        const int c = c0 + tx;
        tile[ty + r * 8][tx] = (p < HW && c < C) ? tb[p * C + c] : 0.f;
    }
    __syncthreads();
    const float a = alpha[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + r * 8;
        const long p = p0 + tx;
        if (c < C && p < HW) {
            const long o = ((long)b * C + c) * HW + p;
            y[o] = a * tile[tx][ty + r * 8] + x[o];
        }
    }
}

size_t fl(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace

extern "C" {

size_t mi355_sk_workspace_bytes(int B, int planes, int H, int W) {
    if (B <= 0 || planes <= 0 || H <= 0 || W <= 0) return 0;
    return 4 * (2 * fl((size_t)B * planes * H * W) + fl((size_t)B * planes) + fl((size_t)B * 2 * planes)) + 256;
}

int mi355_sk_fwd(const float* x, const float* const* p, float* y, int B, int Cin, int planes, int groups, int d, int H, int W,
                 void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && p && y && workspace && B > 0 && Cin > 0 && planes > 0 && groups > 0 && d > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(Cin % groups == 0 && planes % groups == 0);
    for (int q = 0; q < MI355_SK_NPARAMS; ++q) MI355_CHECK_ARG(p[q] != nullptr);
    MI355_CHECK_ARG(workspace_bytes >= mi355_sk_workspace_bytes(B, planes, H, W) && aligned16(workspace));
    const int cog = planes / groups, cin_g = Cin / groups;
    if (cog != 1 && cog != 2 && cog != 4 && cog != 8 && cog != 16)
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_sk_fwd: planes / groups = %d not in {1,2,4,8,16}", cog);
    if (d > 8192) return mi355::fail(MI355_EUNSUPPORTED, "mi355_sk_fwd: d = %d > 8192", d);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* u1 = ws;                                            // split_3x3(x): sk_module.py:43
    float* u2 = u1 + fl((size_t)B * planes * HW);              // split_5x5(x): :44
    float* pooled = u2 + fl((size_t)B * planes * HW);          // sum_hw (u1 + u2): :45-46
    float* att = pooled + fl((size_t)B * planes);              // softmax over the branches: :48-50
    const dim3 grid(groups, B);
#define SKB(COG_, PIX_)                                                                                                             \
    sk_branch_kernel<COG_, PIX_><<<grid, 256, 0, st>>>(x, p[MI355_SK_CONV3_W], p[MI355_SK_CONV3_SCALE], p[MI355_SK_CONV3_SHIFT],   \
                                                       p[MI355_SK_CONV5_W], p[MI355_SK_CONV5_SCALE], p[MI355_SK_CONV5_SHIFT], u1, u2, \
                                                       pooled, Cin, planes, cin_g, H, W)
    switch (cog) {
        case 1: SKB(1, 4); break;
        case 2: SKB(2, 4); break;
        case 4: SKB(4, 4); break;
        case 8: SKB(8, 4); break;
        default: SKB(16, 2); break;
    }
#undef SKB
    sk_select_kernel<<<B, 256, d * sizeof(float), st>>>(pooled, p[MI355_SK_FC_W], p[MI355_SK_FC_B], p[MI355_SK_FC_BN_SCALE], p[MI355_SK_FC_BN_SHIFT],
                                                        p[MI355_SK_FC1_W], p[MI355_SK_FC1_B], p[MI355_SK_FC2_W], p[MI355_SK_FC2_B], att, planes, d,
                                                        1.0f / (float)HW);
    const long n = (long)B * planes * HW;
    if ((HW & 3) == 0 && aligned16(y)) sk_apply_kernel<4><<<cdiv(n / 4, 256), 256, 0, st>>>(u1, u2, att, y, n / 4, HW, planes);
    else                               sk_apply_kernel<1><<<cdiv(n, 256), 256, 0, st>>>(u1, u2, att, y, n, HW, planes);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

size_t mi355_cam_workspace_bytes(int B, int C) { return (B <= 0 || C <= 0) ? 0 : 4 * fl((size_t)B * C * C) + 256; }

int mi355_cam_fwd(const float* x, const float* beta, float* y, int B, int C, int H, int W, int precision, void* workspace,
                  size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && beta && y && workspace && B > 0 && C > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(precision >= MI355_PREC_STRICT && precision <= MI355_PREC_BF16);
    MI355_CHECK_ARG(workspace_bytes >= mi355_cam_workspace_bytes(B, C) && aligned16(workspace));
    const long HW = (long)H * W;
    if ((HW & 3) || (C & 3) || !aligned16(x) || !aligned16(y))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_cam_fwd: H*W and C must be multiples of 4 (HW=%ld C=%d)", HW, C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* G = static_cast<float*>(workspace);                  // x_ x_^T: dual_attention.py:38
    int rc = mi355::gemm_nt_batched(x, x, G, B, C, C, (int)HW, (int)HW, (int)HW, C, (long)C * HW, (long)C * HW, (long)C * C, precision, st);
    if (rc) return rc;
    softmax_rows_scaled_kernel<<<cdiv((long)B * C, 4), 256, 0, st>>>(G, C, (long)B * C, beta);      // softmax: :39, beta: :41
    rc = mi355::gemm_kn_batched(G, x, nullptr, x, y, B, C, (int)HW, C, C, (int)HW, (int)HW, (long)C * C, (long)C * HW, (long)C * HW,
                                MI355_ACT_NONE, precision, st);  // attn @ x_ + x: :40-41
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_tokens_to_nchw_axpy_fwd(const float* tokens, const float* x, const float* alpha, float* y, int B, int HW, int C,
                                  mi355_stream_t stream) {
    MI355_CHECK_ARG(tokens && x && alpha && y && B > 0 && HW > 0 && C > 0 && B <= 65535);
    tokens_to_nchw_axpy_kernel<<<dim3(cdiv(HW, 32), cdiv(C, 32), B), 256, 0, static_cast<hipStream_t>(stream)>>>(tokens, x, alpha, y, HW, C);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
