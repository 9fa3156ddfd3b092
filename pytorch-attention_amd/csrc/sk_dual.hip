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

// The same two branches with the group's input staged in LDS: a workgroup owns (group, band of BR rows, image), parks the cin_g input
// planes of the band with a zero halo of two (rows and columns; row pitch PW = W + 4 rounded to 4) and the group's weights as
// [ci][tap][branch][COG]; a thread computes four horizontally adjacent pixels of both branches from five 8-float row windows
// per input channel (two aligned 16-byte LDS reads each: columns j-2 .. j+5 cover the taps of both dilations).  No bounds checks and
// no global address arithmetic in the FMA loop.  pooled_part[(b, channel, band)] holds the band's share of sum_hw (u1 + u2).
template <int COG>
__global__ __launch_bounds__(256) void sk_branch_lds_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ sc1,
                                                           const float* __restrict__ sh1, const float* __restrict__ w2, const float* __restrict__ sc2,
                                                           const float* __restrict__ sh2, float* __restrict__ u1, float* __restrict__ u2,
                                                           float* __restrict__ pooled_part, int Cin, int planes, int cin_g, int H, int W, int BR,
                                                           int PW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[4][COG];
    const int g = blockIdx.x, band = blockIdx.y, nb = gridDim.y, b = blockIdx.z, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = band * BR, TS = (BR + 4) * PW;                // tile stride per input channel
    float* tile = lds;                                           // cin_g * TS
    float* wl = lds + cin_g * TS;                                // cin_g * 9 * 2 * COG
    const long HW = (long)H * W;
    const float* xg = x + ((long)b * Cin + (long)g * cin_g) * HW;
    const int co0 = g * COG, w4 = W >> 2;
    for (int q = t; q < cin_g * TS; q += 256) tile[q] = 0.f;
    for (int q = t; q < cin_g * 9 * 2 * COG; q += 256) {
        const int o = q % COG, br = (q / COG) & 1, tap = (q / (2 * COG)) % 9, ci = q / (18 * COG);
        wl[q] = (br ? w2 : w1)[((long)(co0 + o) * cin_g + ci) * 9 + tap];
    }
    __syncthreads();
    for (int q = t; q < cin_g * (BR + 4) * w4; q += 256) {
        const int c4 = q % w4, rr = (q / w4) % (BR + 4), ci = q / (w4 * (BR + 4));
        const int i = i0 - 2 + rr;
        if (i >= 0 && i < H) {
            const f4 v = *reinterpret_cast<const f4*>(xg + (long)ci * HW + (long)i * W + c4 * 4);
            float* d = tile + ci * TS + rr * PW + 2 + c4 * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    float psum[COG];
#pragma unroll
    for (int o = 0; o < COG; ++o) psum[o] = 0.f;
    for (int item = t; item < BR * w4; item += 256) {
        const int r = item / w4, c4 = item - r * w4, i = i0 + r;
        if (i >= H) continue;
        float a1[COG][4], a2[COG][4];
#pragma unroll
        for (int o = 0; o < COG; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1[o][e] = 0.f; a2[o][e] = 0.f; }
        for (int ci = 0; ci < cin_g; ++ci) {
            const float* tp = tile + ci * TS + r * PW + c4 * 4;
            float f[5][8];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const f4 lo = *reinterpret_cast<const f4*>(tp + k * PW), hi = *reinterpret_cast<const f4*>(tp + k * PW + 4);
                f[k][0] = lo.x; f[k][1] = lo.y; f[k][2] = lo.z; f[k][3] = lo.w;
                f[k][4] = hi.x; f[k][5] = hi.y; f[k][6] = hi.z; f[k][7] = hi.w;
            }
            const float* wc = wl + ci * 18 * COG;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const float* wt = wc + (u * 3 + v) * 2 * COG;
#pragma unroll
                    for (int o = 0; o < COG; ++o) {
                        const float wa = wt[o], wb = wt[COG + o];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a1[o][e] = __builtin_fmaf(wa, f[1 + u][e + 1 + v], a1[o][e]);        // dilation 1: rows i-1..i+1, cols j-1..j+1
                            a2[o][e] = __builtin_fmaf(wb, f[2 * u][e + 2 * v], a2[o][e]);        // dilation 2: rows i-2..i+2, cols j-2..j+2
                        }
                    }
                }
        }
#pragma unroll
        for (int o = 0; o < COG; ++o) {
            const float s1 = sc1[co0 + o], t1 = sh1[co0 + o], s2 = sc2[co0 + o], t2 = sh2[co0 + o];
            const long off = ((long)b * planes + co0 + o) * HW + (long)i * W + c4 * 4;
            f4 r1, r2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { r1[e] = fmaxf(a1[o][e] * s1 + t1, 0.f); r2[e] = fmaxf(a2[o][e] * s2 + t2, 0.f); }
            *reinterpret_cast<f4*>(u1 + off) = r1;
            *reinterpret_cast<f4*>(u2 + off) = r2;
            psum[o] += ((r1.x + r2.x) + (r1.y + r2.y)) + ((r1.z + r2.z) + (r1.w + r2.w));
        }
    }
#pragma unroll
    for (int o = 0; o < COG; ++o) {
        const float s = wave_sum(psum[o]);
        if (lane == 0) red[wave][o] = s;
    }
    __syncthreads();
    if (t < COG) pooled_part[((long)b * planes + co0 + t) * nb + band] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// fc -> BatchNorm1d (folded) -> ReLU -> fc1 / fc2 -> softmax over the two branches; one workgroup per image.  att: (B, 2, planes)
__global__ __launch_bounds__(256) void sk_select_kernel(const float* __restrict__ pooled, const float* __restrict__ wf, const float* __restrict__ bf,
                                                       const float* __restrict__ bn_s, const float* __restrict__ bn_t, const float* __restrict__ wa,
                                                       const float* __restrict__ ba, const float* __restrict__ wb, const float* __restrict__ bb,
                                                       float* __restrict__ att, int planes, int d, float inv_hw, int nb) {
    extern __shared__ float z[];                                 // d | planes
    float* s = z + d;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    for (int c = t; c < planes; c += 256) {                      // the bands' shares of the pooled sum, added in band order
        const float* pp = pooled + ((long)b * planes + c) * nb;
        float acc = 0.f;
        for (int q = 0; q < nb; ++q) acc += pp[q];
        s[c] = acc * inv_hw;
    }
    __syncthreads();
    for (int k = wave; k < d; k += 4) {
        float acc = 0.f;
        for (int c = lane; c < planes; c += 64) acc = __builtin_fmaf(wf[(long)k * planes + c], s[c], acc);
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
    const float a = alpha ? alpha[0] : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + r * 8;
        const long p = p0 + tx;
        if (c < C && p < HW) {
            const long o = ((long)b * C + c) * HW + p;
            y[o] = a * tile[tx][ty + r * 8] + (x ? x[o] : 0.f);
        }
    }
}

size_t fl(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace

extern "C" {

size_t mi355_sk_workspace_bytes(int B, int planes, int H, int W) {
    if (B <= 0 || planes <= 0 || H <= 0 || W <= 0) return 0;
    return 4 * (2 * fl((size_t)B * planes * H * W) + fl((size_t)B * planes * H) + fl((size_t)B * 2 * planes)) + 256;
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
    if (d + planes > 12288) return mi355::fail(MI355_EUNSUPPORTED, "mi355_sk_fwd: d + planes = %d > 12288", d + planes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* u1 = ws;                                            // split_3x3(x): sk_module.py:43
    float* u2 = u1 + fl((size_t)B * planes * HW);              // split_5x5(x): :44
    float* pooled = u2 + fl((size_t)B * planes * HW);          // sum_hw (u1 + u2): :45-46
    float* att = pooled + fl((size_t)B * planes * H);          // softmax over the branches: :48-50
    // LDS-tiled kernel when W % 4 == 0 and a band of >= 1 rows of the group's input (+ halo) fits 48 KB; bands are balanced
    int nb = 1;
    const int PW = (W + 4 + 3) & ~3;
    const size_t wbytes = (size_t)cin_g * 18 * cog * sizeof(float);
    const size_t row_bytes = (size_t)cin_g * PW * sizeof(float);
    const long max_rows = ((long)49152 - (long)wbytes) / (long)row_bytes - 4;
    const bool tiled = (W & 3) == 0 && aligned16(x) && aligned16(u1) && max_rows >= 1 && cog <= 8 && B <= 65535;
    if (tiled) {
        int brmax = max_rows < H ? (int)max_rows : H;
        const int fill = 256 / (W >> 2);                          // rows that one pass of the 256 threads covers (a thread = 4 pixels)
        if (fill >= 1 && brmax > fill) brmax = fill;
        nb = cdiv(H, brmax);
        const int BR = cdiv(H, nb);
        nb = cdiv(H, BR);
        const size_t lds = (size_t)cin_g * (BR + 4) * PW * sizeof(float) + wbytes;
        const dim3 tgrid(groups, nb, B);
#define SKT(COG_)                                                                                                                    \
    sk_branch_lds_kernel<COG_><<<tgrid, 256, lds, st>>>(x, p[MI355_SK_CONV3_W], p[MI355_SK_CONV3_SCALE], p[MI355_SK_CONV3_SHIFT],    \
                                                        p[MI355_SK_CONV5_W], p[MI355_SK_CONV5_SCALE], p[MI355_SK_CONV5_SHIFT], u1, u2, \
                                                        pooled, Cin, planes, cin_g, H, W, BR, PW)
        switch (cog) {
            case 1: SKT(1); break;
            case 2: SKT(2); break;
            case 4: SKT(4); break;
            default: SKT(8); break;
        }
#undef SKT
    } else {
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
    }
    sk_select_kernel<<<B, 256, (size_t)(d + planes) * sizeof(float), st>>>(pooled, p[MI355_SK_FC_W], p[MI355_SK_FC_B], p[MI355_SK_FC_BN_SCALE],
                                                                          p[MI355_SK_FC_BN_SHIFT], p[MI355_SK_FC1_W], p[MI355_SK_FC1_B], p[MI355_SK_FC2_W],
                                                                          p[MI355_SK_FC2_B], att, planes, d, 1.0f / (float)HW, nb);
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
    // the logits are unscaled sums over HW (O(HW) on the diagonal): always the fp32-class split-bf16 mode; `precision` selects the
    // operand format of the second product only (softmax weights in [0, 1], residual added in fp32)
    int rc = mi355::gemm_nt_batched(x, x, G, B, C, C, (int)HW, (int)HW, (int)HW, C, (long)C * HW, (long)C * HW, (long)C * C,
                                    MI355_PREC_STRICT, st);
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
    MI355_CHECK_ARG(tokens && y && B > 0 && HW > 0 && C > 0 && B <= 65535);
    tokens_to_nchw_axpy_kernel<<<dim3(cdiv(HW, 32), cdiv(C, 32), B), 256, 0, static_cast<hipStream_t>(stream)>>>(tokens, x, alpha, y, HW, C);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
