// xcit.hip -- XCiT cross-covariance attention core (XCA) and the LPI depth-wise stencil block for gfx950.
//
// XCA (xcit.py:249-262): per (image, head) the attention matrix is only d x d (48 x 48 for XCiT-S) and both
// contractions run over <= 224 tokens, so the whole head lives in one workgroup's LDS and the arithmetic is done in
// EXACT fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, cdna_hip_programming.md section 3) --
// no operand rounding, hence the `precision` argument does not change the result:
//     G = Q^T K (d x d, contraction over tokens)   ->  G_ij / (max(|q_i|,eps) max(|k_j|,eps)) * temperature_h
//     A = softmax_rows(G)                          ->  O = A V^T (d x N)  ->  out[b, n, h*d + i] = O[i][n]
// HBM traffic = read the head's q,k,v once + write out once.
//
// LPI (xcit.py:149-157): tokens -> (C,H,W) image -> dw3x3 -> GELU -> BatchNorm(eval) -> dw3x3 -> tokens, fused in one
// kernel per (image, 32-channel group): the token tile and the intermediate sit in LDS, lanes run along channels so
// every LDS access is conflict-free and every HBM access is a 128-byte row piece; LayerScale + residual fused.
#include "common.h"
#include "mma.h"

namespace {

constexpr int XCA_NP = 224;     // max tokens (multiple of 16)

template <int D>
__global__ __launch_bounds__(256) void xca_kernel(const float* __restrict__ qkv, const float* __restrict__ temperature,
                                                 float* __restrict__ out, int N, int heads) {
    constexpr int P = D + 1;                 // LDS pitch (floats): odd -> conflict-free column walks
    constexpr int DT = D / 16;
    __shared__ float s_a[XCA_NP * P];        // q, later v
    __shared__ float s_b[XCA_NP * P];        // k
    __shared__ float s_g[D * P];             // G, then A = softmax(G)
    __shared__ float s_n[2 * D];             // column norms of q and k
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % heads, b = blockIdx.x / heads;
    const int C = heads * D;
    const long row3 = 3L * C;
    const float* base = qkv + (long)b * N * row3 + h * D;
    const int NP4 = (N + 3) & ~3, NP16 = (N + 15) & ~15;

    // ---- stage q, k (rows >= N zeroed up to the 16-token boundary) ---------------------------------------------------
    constexpr int D4 = D / 4;
    for (int idx = t; idx < NP16 * D4; idx += 256) {
        const int n = idx / D4, d4 = idx % D4;
        f4 q = {0.f, 0.f, 0.f, 0.f}, k = {0.f, 0.f, 0.f, 0.f};
        if (n < N) {
            q = *reinterpret_cast<const f4*>(base + (long)n * row3 + d4 * 4);
            k = *reinterpret_cast<const f4*>(base + (long)n * row3 + C + d4 * 4);
        }
        float* pa = s_a + n * P + d4 * 4;
        float* pb = s_b + n * P + d4 * 4;
        pa[0] = q.x; pa[1] = q.y; pa[2] = q.z; pa[3] = q.w;
        pb[0] = k.x; pb[1] = k.y; pb[2] = k.z; pb[3] = k.w;
    }
    __syncthreads();
    // ---- column L2 norms over tokens (F.normalize: x / max(||x||, 1e-12)) --------------------------------------------
    if (t < 2 * D) {
        const float* src = (t < D ? s_a : s_b) + (t % D);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int n = 0;
        for (; n + 3 < N; n += 4) {
            const float v0 = src[n * P], v1 = src[(n + 1) * P], v2 = src[(n + 2) * P], v3 = src[(n + 3) * P];
            a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
        }
        for (; n < N; ++n) { const float v = src[n * P]; a0 += v * v; }
        s_n[t] = fmaxf(sqrtf((a0 + a1) + (a2 + a3)), 1e-12f);
    }
    // ---- G = Q^T K on exact-fp32 MFMA: A[i][k=n] = q[n][i], B[k=n][j] = k[n][j] ------------------------------------------
    for (int tl = wave; tl < DT * DT; tl += 4) {
        const int it = tl / DT, jt = tl % DT;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < NP4 / 4; ++ks) {
            const float av = s_a[(ks * 4 + g) * P + it * 16 + l15];
            const float bv = s_b[(ks * 4 + g) * P + jt * 16 + l15];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_g[(it * 16 + g * 4 + r) * P + jt * 16 + l15] = acc[r];
    }
    __syncthreads();
    // ---- A = softmax_rows(G / (|q_i| |k_j|) * temperature): 16 lanes per row ------------------------------------------
    {
        const float temp = temperature[h];
        const int tj = t & 15;
        for (int i = t >> 4; i < D; i += 16) {
            float v[DT];
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                const int j = tj + 16 * c;
                v[c] = s_g[i * P + j] / (s_n[i] * s_n[D + j]) * temp;
                m = fmaxf(m, v[c]);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < DT; ++c) { v[c] = expf(v[c] - m); sum += v[c]; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, WAVE);
#pragma unroll
            for (int c = 0; c < DT; ++c) s_g[i * P + tj + 16 * c] = v[c] / sum;
        }
    }
    // ---- stage v over q's buffer -----------------------------------------------------------------------------------------
    for (int idx = t; idx < NP16 * D4; idx += 256) {
        const int n = idx / D4, d4 = idx % D4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *reinterpret_cast<const f4*>(base + (long)n * row3 + 2 * C + d4 * 4);
        float* pa = s_a + n * P + d4 * 4;
        pa[0] = v.x; pa[1] = v.y; pa[2] = v.z; pa[3] = v.w;
    }
    __syncthreads();
    // ---- O[i][n] = sum_j A[i][j] v[n][j]:  A-operand rows i, B-operand columns n ------------------------------------------
    const int NT = NP16 / 16;
    for (int tl = wave; tl < DT * NT; tl += 4) {
        const int it = tl % DT, nt = tl / DT;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D / 4; ++ks) {
            const float av = s_g[(it * 16 + l15) * P + ks * 4 + g];
            const float bv = s_a[(nt * 16 + l15) * P + ks * 4 + g];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        const int n = nt * 16 + l15;
        if (n < N) *reinterpret_cast<f4*>(out + ((long)b * N + n) * C + h * D + it * 16 + g * 4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LPI: workgroup = (image, 32 channels); 256 threads = 8 token lanes x 32 channel lanes.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int LPI_CG = 32;

__global__ __launch_bounds__(256) void lpi_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ bn_w,
                                                 const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                                 const float* __restrict__ bn_v, float bn_eps, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, const float* __restrict__ gamma,
                                                 const float* __restrict__ resid, float* __restrict__ y, int H, int W, int C,
                                                 int groups) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = H * W;
    float* s_x = smem;                 // [N][32]
    float* s_m = smem + N * LPI_CG;    // [N][32] intermediate
    const int t = threadIdx.x, cl = t & 31, tl = t >> 5;
    const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * LPI_CG + cl;
    const bool cok = c < C;
    const float* xb = x + (long)b * N * C;
    for (int n = tl; n < N; n += 8) s_x[n * LPI_CG + cl] = cok ? xb[(long)n * C + c] : 0.f;
    float k1[9], k2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { k1[i] = cok ? w1[(long)c * 9 + i] : 0.f; k2[i] = cok ? w2[(long)c * 9 + i] : 0.f; }
    const float bias1 = cok ? b1[c] : 0.f, bias2 = cok ? b2[c] : 0.f;
    const float mean = cok ? bn_m[c] : 0.f, var = cok ? bn_v[c] : 1.f, bw = cok ? bn_w[c] : 0.f, bb = cok ? bn_b[c] : 0.f;
    const float rstd = 1.0f / sqrtf(var + bn_eps);
    __syncthreads();
    auto conv = [&](const float* src, const float* k, float bias, int n) {
        const int yy = n / W, xx = n % W;
        float acc = bias;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int y2 = yy + dy, x2 = xx + dx;
                if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) acc += k[(dy + 1) * 3 + dx + 1] * src[(y2 * W + x2) * LPI_CG + cl];
            }
        return acc;
    };
    for (int n = tl; n < N; n += 8) {
        const float v = gelu_erf(conv(s_x, k1, bias1, n));
        s_m[n * LPI_CG + cl] = (v - mean) * rstd * bw + bb;
    }
    __syncthreads();
    if (!cok) return;
    const float gm = gamma ? gamma[c] : 1.0f;
    for (int n = tl; n < N; n += 8) {
        float v = conv(s_m, k2, bias2, n);
        const long o = ((long)b * N + n) * C + c;
        v = gamma ? v * gm : v;
        if (resid) v += resid[o];
        y[o] = v;
    }
}

}  // namespace

extern "C" {

int mi355_xca_fwd(const float* qkv, const float* temperature, float* out, int B, int N, int heads, int d, int precision,
                  mi355_stream_t stream) {
    MI355_CHECK_ARG(qkv && temperature && out && B > 0 && N > 0 && heads > 0);
    MI355_CHECK_ARG(precision >= 0 && precision <= 2);      // accepted for API symmetry; the core is exact fp32 in every mode
    MI355_CHECK_ARG(aligned16(qkv) && aligned16(out));
    if (N > XCA_NP) return mi355::fail(MI355_EUNSUPPORTED, "mi355_xca_fwd: %d tokens > %d (one head must fit one workgroup's LDS)", N, XCA_NP);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = B * heads;
    switch (d) {
        case 32: xca_kernel<32><<<grid, 256, 0, st>>>(qkv, temperature, out, N, heads); break;
        case 48: xca_kernel<48><<<grid, 256, 0, st>>>(qkv, temperature, out, N, heads); break;
        case 64:
            
            xca_kernel<64><<<grid, 256, 0, st>>>(qkv, temperature, out, N, heads); break;
        default: return mi355::fail(MI355_EUNSUPPORTED, "mi355_xca_fwd: head dim %d (built: 32, 48, 64)", d);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

size_t mi355_lpi_workspace_bytes(int, int, int, int) { return 16; }   // fully fused: no scratch needed

int mi355_lpi_fwd(const float* x, const float* w1, const float* b1, const float* bn_w, const float* bn_b, const float* bn_mean,
                  const float* bn_var, float bn_eps, const float* w2, const float* b2, const float* gamma, const float* resid,
                  float* y, int B, int H, int W, int C, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w1 && b1 && bn_w && bn_b && bn_mean && bn_var && w2 && b2 && y);
    MI355_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0);
    (void)ws; (void)ws_bytes;
    const size_t smem = (size_t)2 * H * W * LPI_CG * sizeof(float);
    if (smem > 64 * 1024) return mi355::fail(MI355_EUNSUPPORTED, "mi355_lpi_fwd: %dx%d token grid exceeds the LDS tile (<= 256 tokens)", H, W);
    const int groups = cdiv(C, LPI_CG);
    lpi_kernel<<<B * groups, 256, smem, static_cast<hipStream_t>(stream)>>>(x, w1, b1, bn_w, bn_b, bn_mean, bn_var, bn_eps, w2, b2,
                                                                            gamma, resid, y, H, W, C, groups);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
