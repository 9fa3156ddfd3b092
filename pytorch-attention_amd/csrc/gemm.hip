// gemm.hip -- MFMA GEMM with fused epilogue for gfx950: nn.Linear, Mixer token mixing, ViT patch embedding.
//
//   C[b] (M x N) = epilogue( A[b] (M x K) . B[b] )       fp32 in HBM, 16-bit MFMA operands, fp32 accumulate
//
// Operand layouts (row-major, strides in elements):
//   A      : (M, K), K contiguous                      AMODE 0 plain rows | AMODE 1 ViT patch gather (im2col in the load)
//   B NT   : (N, K), K contiguous  (nn.Linear weight)  BMODE 0
//   B KN   : (K, N), N contiguous  (activation as the K-major operand: Mixer token mix, 1x1 conv on NCHW)  BMODE 1
// Tiling: 128 x 128 x 32 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles).
// Staging: global fp32 -> registers (float4, issued one K-step ahead) -> convert to the MFMA operand format (hi/lo
// pair in strict mode) -> LDS rows of 32 k-elements (+8 pad), always K-contiguous; the KN operand is transposed in
// registers (4 k-rows x 4 n-columns micro-tile per thread) on its way to LDS.  Two LDS buffers, one barrier per K-step.
// Epilogue: accumulators -> per-wave LDS slab -> row-contiguous float4 stores with bias / GELU / LayerScale /
// residual / position-embedding fused (the C/D fragment layout would otherwise write 64-byte pieces).
// Workgroup ids are remapped so that each XCD owns a contiguous run of tiles (n fastest): the A panel of a tile row
// is fetched into one L2 instead of eight.
#include "common.h"
#include "mma.h"
#include "gemm16.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK + 8;                 // LDS row pitch in 16-bit elements (80 B, keeps 16-B alignment)
constexpr int EPITCH = 68;                    // epilogue slab pitch in floats

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* gamma; const float* resid; const float* pos;
    int M, N, K;
    int lda, ldb, ldc;
    long sA, sB, sC;                          // batch strides (elements); 0 = shared operand
    int act, bias_per_row;
    // AMODE 1 (patch embedding) / 2, 3 (convolution as implicit GEMM): image geometry
    int Cin, H, W, ps, gw, P;
    int Pout;                                 // output rows per image (P + 1 when a cls row follows the patches, else P)
    int KH, KW, stride, pad, OW, Kreal;       // AMODE 2 / 3 (Kreal = Cin*KH*KW; K may be padded up to a multiple of 4)
};

__device__ __forceinline__ f4 ld4_guard(const float* p, bool ok) {
    f4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? *reinterpret_cast<const f4*>(p) : z;
}

template <int PREC>
__device__ __forceinline__ void st_lds4(unsigned short* hi, unsigned short* lo, int off, f4 v) {
    using M_ = Mma<PREC>;
    typename M_::v4 h = M_::cvt(v);
    *reinterpret_cast<typename M_::v4*>(hi + off) = h;
    if constexpr (M_::NSPLIT == 2) *reinterpret_cast<typename M_::v4*>(lo + off) = M_::cvt_lo(v, h);
}

template <int PREC, int BMODE, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    constexpr int NS = M_::NSPLIT;
    constexpr int TILE = (BM + BN) * PITCH;                  // 16-bit elements per (split, buffer)
    constexpr int STAGE_BYTES = 2 * NS * TILE * 2;
    constexpr int EPI_BYTES = 4 * 32 * EPITCH * 4;
    constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned short* lds = reinterpret_cast<unsigned short*>(lds_raw);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- tile id: XCD-contiguous remap (bijective for any grid size), n fastest -------------------------
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int bz = blockIdx.y;
    const float* __restrict__ Ab = g.A + (long)bz * g.sA;
    const float* __restrict__ Bb = g.B + (long)bz * g.sB;

    // ---- per-thread staging coordinates --------------------------------------------------------------------
    const int lr = t >> 3, lk = (t & 7) * 4;                 // row-within-32 and k offset for K-contiguous operands
    const float* a_ptr[4];
    bool a_ok[4];
    int a_iy[4] = {0, 0, 0, 0}, a_ix[4] = {0, 0, 0, 0};      // AMODE 2 / 3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lr + 32 * i;
        a_ok[i] = m < g.M;
        if constexpr (AMODE == 0) {
            a_ptr[i] = Ab + (long)(a_ok[i] ? m : 0) * g.lda;
        } else if constexpr (AMODE == 1) {                   // row m = (image, patch); k = (c, ky, kx), kx contiguous
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / g.P, p = mm % g.P;
            const int py = p / g.gw, px = p % g.gw;
            a_ptr[i] = Ab + ((long)img * g.Cin * g.H + (long)py * g.ps) * g.W + (long)px * g.ps;
        } else {                                             // convolution: row m = (image, oy, ox); top-left input coordinate
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / g.P, p = mm % g.P;
            a_iy[i] = (p / g.OW) * g.stride - g.pad;
            a_ix[i] = (p % g.OW) * g.stride - g.pad;
            a_ptr[i] = Ab + (long)img * g.Cin * g.H * g.W;   // image base (NCHW and token-major NHWC have the same size)
        }
    }
    const float* b_ptr[4];
    bool b_ok[4];
    const int bk = (t & 7) * 4, bn4 = (t >> 3) * 4;          // BMODE 1: k-group and n-quad of this thread's 4x4 micro-tile
    if constexpr (BMODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + lr + 32 * i;
            b_ok[i] = n < g.N;
            b_ptr[i] = Bb + (long)(b_ok[i] ? n : 0) * g.ldb;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b_ok[i] = (n0 + bn4) < g.N;
            b_ptr[i] = Bb + n0 + bn4;                        // + k * ldb added per step
        }
    }

    f4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + lk;
            if constexpr (AMODE == 0) {
                ra[i] = ld4_guard(a_ptr[i] + k, a_ok[i] && k < g.K);
            } else if constexpr (AMODE == 1) {
                const int pp = g.ps * g.ps;
                const int c = k / pp, rem = k % pp;
                const int ky = rem / g.ps, kx = rem % g.ps;
                ra[i] = ld4_guard(a_ptr[i] + ((long)c * g.H + ky) * g.W + kx, a_ok[i] && k < g.K);
            } else if constexpr (AMODE == 2) {               // NCHW input, k = (c, ky, kx): element-wise gather, zero padding
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = k + q, kk2 = g.KH * g.KW;
                    const int c = kq / kk2, rem = kq % kk2;
                    const int iy = a_iy[i] + rem / g.KW, ix = a_ix[i] + rem % g.KW;
                    const bool ok = a_ok[i] && kq < g.Kreal && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    e[q] = ok ? a_ptr[i][((long)c * g.H + iy) * g.W + ix] : 0.f;
                }
                ra[i] = f4{e[0], e[1], e[2], e[3]};
            } else {                                         // token-major (H*W, C) input, k = (ky, kx, c): 4 channels per load
                const int tap = k / g.Cin, c = k % g.Cin;
                const int iy = a_iy[i] + tap / g.KW, ix = a_ix[i] + tap % g.KW;
                const bool ok = a_ok[i] && k < g.K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                ra[i] = ld4_guard(a_ptr[i] + ((long)iy * g.W + ix) * g.Cin + c, ok);
            }
        }
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + lk;
                rb[i] = ld4_guard(b_ptr[i] + k, b_ok[i] && k < g.K);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + bk + j;
                rb[j] = ld4_guard(b_ptr[j] + (long)k * g.ldb, b_ok[j] && k < g.K);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned short* hiA = lds + (buf * NS + 0) * TILE;
        unsigned short* loA = lds + (buf * NS + (NS - 1)) * TILE;
        unsigned short* hiB = hiA + BM * PITCH;
        unsigned short* loB = loA + BM * PITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) st_lds4<PREC>(hiA, loA, (lr + 32 * i) * PITCH + lk, ra[i]);
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st_lds4<PREC>(hiB, loB, (lr + 32 * i) * PITCH + lk, rb[i]);
        } else {                                             // transpose the 4(k) x 4(n) micro-tile: rows n, 4 consecutive k
            const f4 c0 = {rb[0].x, rb[1].x, rb[2].x, rb[3].x};
            const f4 c1 = {rb[0].y, rb[1].y, rb[2].y, rb[3].y};
            const f4 c2 = {rb[0].z, rb[1].z, rb[2].z, rb[3].z};
            const f4 c3 = {rb[0].w, rb[1].w, rb[2].w, rb[3].w};
            st_lds4<PREC>(hiB, loB, (bn4 + 0) * PITCH + bk, c0);
            st_lds4<PREC>(hiB, loB, (bn4 + 1) * PITCH + bk, c1);
            st_lds4<PREC>(hiB, loB, (bn4 + 2) * PITCH + bk, c2);
            st_lds4<PREC>(hiB, loB, (bn4 + 3) * PITCH + bk, c3);
        }
    };

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);           // next tile in flight under the MFMAs
        const unsigned short* sA = lds + (buf * NS) * TILE + (wr * 64 + frow) * PITCH + fk;
        const unsigned short* sB = lds + (buf * NS) * TILE + BM * PITCH + (wc * 64 + frow) * PITCH + fk;
        v8 fa[4][NS], fb[4][NS];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fa[i][s] = *reinterpret_cast<const v8*>(sA + s * TILE + i * 16 * PITCH);
                fb[i][s] = *reinterpret_cast<const v8*>(sB + s * TILE + i * 16 * PITCH);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma_step<PREC>(fa[i], fb[j], acc[i][j]);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: 2 passes of 32 rows per wave through a private LDS slab ---------------------------------
    float* slab = reinterpret_cast<float*>(lds_raw) + wave * 32 * EPITCH;
    float* Cb = g.C + (long)bz * g.sC;
    const float* Rb = g.resid ? g.resid + (long)bz * g.sC : nullptr;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0) &&
                        (!Rb || (reinterpret_cast<uintptr_t>(Rb) & 15) == 0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[(ii * 16 + (lane >> 4) * 4 + r) * EPITCH + j * 16 + (lane & 15)] = acc[p * 2 + ii][j][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + (lane >> 4), cl = (lane & 15) * 4;
            const int m = m0 + wr * 64 + p * 32 + rl, n = n0 + wc * 64 + cl;
            if (m >= g.M || n >= g.N) continue;
            f4 v = *reinterpret_cast<const f4*>(slab + rl * EPITCH + cl);
            long orow = m;
            if constexpr (AMODE == 1) orow = (long)(m / g.P) * g.Pout + (m % g.P);
            const bool full = vec_ok && (n + 3 < g.N);
            float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (n + c >= g.N) break;
                float z = vv[c];
                if (g.bias) z += g.bias_per_row ? g.bias[m] : g.bias[n + c];
                if constexpr (AMODE >= 1) { if (g.pos) z += g.pos[(long)(m % g.P) * g.N + n + c]; }
                if (g.act == MI355_ACT_GELU) z = gelu_fast(z);
                else if (g.act == MI355_ACT_RELU) z = relu_nan(z);
                if (g.gamma) z *= g.gamma[n + c];
                if (Rb) z += Rb[orow * g.ldc + n + c];
                vv[c] = z;
            }
            if (full) {
                *reinterpret_cast<f4*>(Cb + orow * g.ldc + n) = f4{vv[0], vv[1], vv[2], vv[3]};
            } else {
                for (int c = 0; c < 4 && n + c < g.N; ++c) Cb[orow * g.ldc + n + c] = vv[c];
            }
        }
        __syncthreads();
    }
}

// cls row of the ViT token matrix: tokens[b, P, :] = cls + pos[P]   (ViT.py:183-185, cls appended LAST)
__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ tokens,
                               int B, int P, int E) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * E) return;
    const int b = (int)(i / E), e = (int)(i % E);
    tokens[((long)b * (P + 1) + P) * E + e] = cls[e] + pos[(long)P * E + e];
}

// ---- patch embedding on the 16-bit engine (mi355_patch_embed_ws_fwd) ------------------------------------------------------------------
// im2col in the operand format: block = (image b, patch row py, channel c); the ps image rows of that band (W floats each) are read as
// whole rows into LDS and leave as the (c, ky, kx) slices of the band's W / ps token rows: ps * ps * 2 bytes contiguous per token.
// With a cls row (ViT, cls LAST: row P of every image) that row of the operand is zero, so the product there is bias + table row P.
template <typename T>
__global__ __launch_bounds__(256) void im2col16_kernel(const float* __restrict__ img, T* __restrict__ a16, int Cin, int H, int W, int ps,
                                                       int rows_per_img, int K) {
    extern __shared__ __attribute__((aligned(16))) float s_band[];         // ps x W
    typedef T v8t __attribute__((ext_vector_type(8)));
    const int gw = W / ps, gh = H / ps;
    const int c = blockIdx.x % Cin, py = (blockIdx.x / Cin) % gh, b = blockIdx.x / (Cin * gh);
    const float* src = img + (((long)b * Cin + c) * H + (long)py * ps) * W;
    for (int i = threadIdx.x; i < ps * W / 4; i += 256) reinterpret_cast<f4*>(s_band)[i] = reinterpret_cast<const f4*>(src)[i];
    __syncthreads();
    const int cpr = ps / 8, cpp = ps * cpr;                                  // 16-byte chunks per patch row / per patch
    for (int q = threadIdx.x; q < gw * cpp; q += 256) {
        const int px = q / cpp, r = q - px * cpp, ky = r / cpr, kx0 = (r - ky * cpr) * 8;
        const float* s = s_band + ky * W + px * ps + kx0;
        v8t o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (T)s[k];
        *reinterpret_cast<v8t*>(a16 + ((long)b * rows_per_img + py * gw + px) * K + (long)c * ps * ps + ky * ps + kx0) = o;
    }
    if (py == 0 && rows_per_img > gw * gh) {                                 // the zero row of the cls token
        v8t z;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (T)0.f;
        for (int q = threadIdx.x; q < ps * ps / 8; q += 256)
            *reinterpret_cast<v8t*>(a16 + ((long)b * rows_per_img + gw * gh) * K + (long)c * ps * ps + q * 8) = z;
    }
}

// residual table of the product: rows p < P = pos[p], row P = cls + pos[P] - bias (the GEMM adds the bias to every row)
__global__ void patch_table_kernel(const float* __restrict__ cls, const float* __restrict__ pos, const float* __restrict__ bias,
                                   float* __restrict__ table, int P, int E) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(P + 1) * E) return;
    const int r = (int)(i / E), e = (int)(i % E);
    table[i] = r < P ? pos[i] : cls[e] + pos[i] - bias[e];
}

inline size_t r256(size_t n) { return (n + 255) & ~(size_t)255; }

template <int BMODE, int AMODE>
int launch(const GemmArgs& g, int batch, int precision, hipStream_t st) {
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    dim3 grid(tiles, batch);
    MI355_TRACE(st, "gemm_kernel<prec %d,b%d,a%d> batch=%d M=%d N=%d K=%d", precision, BMODE, AMODE, batch, g.M, g.N, g.K);
    switch (precision) {
        case MI355_PREC_STRICT: gemm_kernel<0, BMODE, AMODE><<<grid, 256, 0, st>>>(g); break;
        case MI355_PREC_FP16:   gemm_kernel<1, BMODE, AMODE><<<grid, 256, 0, st>>>(g); break;
        case MI355_PREC_BF16:   gemm_kernel<2, BMODE, AMODE><<<grid, 256, 0, st>>>(g); break;
        default: return mi355::fail(MI355_EINVAL, "precision must be 0, 1 or 2 (got %d)", precision);
    }
    return MI355_OK;
}

}  // namespace

namespace mi355 {
// Shared with the other translation units (double attention, attention blocks): plain / batched GEMM launches.
int gemm_nt(const float* A, const float* B, const float* bias, const float* gamma, const float* resid, float* C, int M, int N,
            int K, int lda, int ldb, int ldc, int act, int precision, hipStream_t st) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.gamma = gamma; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.act = act;
    return launch<0, 0>(g, 1, precision, st);
}
int gemm_nt_batched(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc, long sA,
                    long sB, long sC, int precision, hipStream_t st) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
    return launch<0, 0>(g, batch, precision, st);
}
int gemm_kn_batched(const float* A, const float* B, const float* bias_row, const float* resid, float* C, int batch, int M,
                    int N, int K, int lda, int ldb, int ldc, long sA, long sB, long sC, int act, int precision,
                    hipStream_t st) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias_row; g.resid = resid; g.bias_per_row = 1;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC; g.act = act;
    return launch<1, 0>(g, batch, precision, st);
}
}  // namespace mi355

extern "C" {

int mi355_linear_fwd(const float* X, const float* W, const float* bias, const float* gamma, const float* resid, float* Y,
                     int M, int N, int K, int ldx, int ldy, int act, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(X && W && Y);
    MI355_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU || act == MI355_ACT_RELU);
    if ((K & 3) || (ldx & 3) || !aligned16(X) || !aligned16(W))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear_fwd: K and ldx must be multiples of 4 and X, W 16-byte aligned "
                                               "(K=%d ldx=%d)", K, ldx);
    int rc = MI355_EUNSUPPORTED;
    if (!gamma && !resid && act == MI355_ACT_NONE)          // small outputs (a classifier head): one-wave 16 x 32 tiles over all CUs, same bits
        rc = mi355::gemm_small_nt(X, W, bias, Y, M, N, K, ldx, K, ldy, precision, static_cast<hipStream_t>(stream));
    if (rc == MI355_EUNSUPPORTED)
        rc = mi355::gemm_nt(X, W, bias, gamma, resid, Y, M, N, K, ldx, K, ldy, act, precision, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_token_mix_fwd(const float* W, const float* X, const float* bias, const float* resid, float* Y, int B, int T, int N,
                        int C, int act, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(W && X && Y);
    MI355_CHECK_ARG(B > 0 && T > 0 && N > 0 && C > 0);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    if ((N & 3) || (C & 3) || !aligned16(X) || !aligned16(W))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_token_mix_fwd: N and C must be multiples of 4 (N=%d C=%d)", N, C);
    int rc = mi355::gemm_kn_batched(W, X, bias, resid, Y, B, T, C, N, N, C, C, 0, (long)N * C, (long)T * C, act, precision,
                                    static_cast<hipStream_t>(stream));
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_patch_embed_fwd(const float* img, const float* Wp, const float* bp, const float* cls, const float* pos, float* tokens,
                          int B, int Cin, int H, int W, int ps, int E, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(img && Wp && bp && tokens);
    MI355_CHECK_ARG((cls == nullptr) == (pos == nullptr));     // both (ViT: cls row + position embedding) or neither (plain patches)
    MI355_CHECK_ARG(B > 0 && Cin > 0 && ps > 0 && E > 0 && H >= ps && W >= ps && H % ps == 0 && W % ps == 0);
    if ((ps & 3) || (W & 3) || !aligned16(img) || !aligned16(Wp))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_patch_embed_fwd: patch size and image width must be multiples of 4");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmArgs g{};
    g.gw = W / ps;
    g.P = (H / ps) * g.gw;
    g.A = img; g.B = Wp; g.C = tokens; g.bias = bp; g.pos = pos;
    g.M = B * g.P; g.N = E; g.K = Cin * ps * ps; g.lda = 0; g.ldb = g.K; g.ldc = E;
    g.Cin = Cin; g.H = H; g.W = W; g.ps = ps;
    g.Pout = cls ? g.P + 1 : g.P;
    int rc = launch<0, 1>(g, 1, precision, st);
    if (rc) return rc;
    if (cls) {
        const long n = (long)B * E;
        cls_row_kernel<<<cdiv(n, 256), 256, 0, st>>>(cls, pos, tokens, B, g.P, E);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// The same through the 16-bit engine when it pays: 16-bit operand modes, ViT form (cls + pos), K = Cin*ps*ps a multiple of 64, ps and E
// multiples of 8.  im2col in the operand format (one pass over the image), then the
// persistent GEMM with the position rows as a periodic residual table -- 0.25 ms -> 0.15 ms at ViT-Base/16, B = 256.  Everything
// else, and any call with too small a workspace, takes mi355_patch_embed_fwd's implicit-GEMM kernel.
static bool patch_embed_fast_ok(const float* cls, int B, int Cin, int H, int W, int ps, int E, int precision) {
    if (!(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16) || cls == nullptr) return false;
    const int K = Cin * ps * ps, P = (H / ps) * (W / ps);
    (void)B;                    // any batch: every 16-bit GEMM kernel keeps a row's K order, so an image's bits do not depend on B
    return !((K % 64) || K < 192 || (ps & 7) || (E & 7) || (W & 3) || P + 1 < 128 || (size_t)ps * W * 4 > 60000);
}

size_t mi355_patch_embed_workspace_bytes(int B, int Cin, int H, int W, int ps, int E, int precision) {
    if (B <= 0 || Cin <= 0 || ps <= 0 || E <= 0 || H < ps || W < ps) return 0;
    static const float dummy = 0.f;
    if (!patch_embed_fast_ok(&dummy, B, Cin, H, W, ps, E, precision)) return 0;
    const size_t K = (size_t)Cin * ps * ps, P = (size_t)(H / ps) * (W / ps);
    return r256((size_t)B * (P + 1) * K * 2) + r256((size_t)E * K * 2) + r256((P + 1) * E * 4);
}

int mi355_patch_embed_ws_fwd(const float* img, const float* Wp, const float* bp, const float* cls, const float* pos, float* tokens,
                             int B, int Cin, int H, int W, int ps, int E, int precision, void* ws, size_t ws_bytes,
                             mi355_stream_t stream) {
    MI355_CHECK_ARG(img && Wp && bp && tokens);
    MI355_CHECK_ARG((cls == nullptr) == (pos == nullptr));
    MI355_CHECK_ARG(B > 0 && Cin > 0 && ps > 0 && E > 0 && H >= ps && W >= ps && H % ps == 0 && W % ps == 0);
    const size_t need = cls ? mi355_patch_embed_workspace_bytes(B, Cin, H, W, ps, E, precision) : 0;
    if (need == 0 || ws == nullptr || ws_bytes < need || !aligned16(ws) || !aligned16(img) || !aligned16(Wp) || !aligned16(tokens) ||
        !aligned16(pos) || !aligned16(bp))
        return mi355_patch_embed_fwd(img, Wp, bp, cls, pos, tokens, B, Cin, H, W, ps, E, precision, stream);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = Cin * ps * ps, P = (H / ps) * (W / ps);
    char* p = static_cast<char*>(ws);
    void* a16 = p;  p += r256((size_t)B * (P + 1) * K * 2);
    void* w16 = p;  p += r256((size_t)E * K * 2);
    float* table = reinterpret_cast<float*>(p);
    int rc = mi355_cast16_fwd(Wp, w16, (size_t)E * K, precision, stream);
    if (rc) return rc;
    {
        MI355_TRACE(st, "patch_table_kernel rows=%d E=%d", P + 1, E);
        patch_table_kernel<<<cdiv((long)(P + 1) * E, 256), 256, 0, st>>>(cls, pos, bp, table, P, E);
    }
    const int blocks = B * (H / ps) * Cin;
    const size_t shm = (size_t)ps * W * 4;
    {
        MI355_TRACE(st, "im2col16_kernel B=%d %dx%d ps=%d", B, H, W, ps);
        if (precision == MI355_PREC_FP16) im2col16_kernel<_Float16><<<blocks, 256, shm, st>>>(img, static_cast<_Float16*>(a16), Cin, H, W, ps, P + 1, K);
        else                              im2col16_kernel<__bf16><<<blocks, 256, shm, st>>>(img, static_cast<__bf16*>(a16), Cin, H, W, ps, P + 1, K);
    }
    g16::G16Args g{};
    g.A = a16; g.B = w16; g.C = tokens; g.bias = bp; g.resid = table; g.resid_period = P + 1;
    g.M = B * (P + 1); g.N = E; g.K = K; g.lda = K; g.ldb = K; g.ldc = E; g.act = MI355_ACT_NONE;
    rc = mi355::linear16_dispatch(g, 0, precision, nullptr, 0, st);
    if (rc == MI355_EUNSUPPORTED) return mi355_patch_embed_fwd(img, Wp, bp, cls, pos, tokens, B, Cin, H, W, ps, E, precision, stream);
    return rc;
}

// Conv2d as implicit GEMM (no im2col buffer): rows = output pixels, K = Cin*KH*KW gathered in the A-operand load with zero padding.
//   in_layout 0: x is NCHW (B,Cin,H,W),            weight rows in (c,ky,kx) order   -- CSWin stem (cswin.py:247-251)
//   in_layout 1: x is token-major (B, H*W, Cin),   weight rows in (ky,kx,c) order   -- CSWin Merge_Block (:218-233)
// weight is (Cout, ldw) with ldw >= Cin*KH*KW, ldw % 4 == 0 and ZERO padding beyond Cin*KH*KW.  y is token-major
// (B, OH*OW, Cout) fp32, + bias.
int mi355_conv2d_tokens_fwd(const float* x, const float* weight, const float* bias, const float* pos, float* y, int B, int Cin, int H,
                            int W, int Cout, int KH, int KW, int stride, int pad, int ldw, int in_layout, int act, int precision,
                            mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && y && B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    MI355_CHECK_ARG(in_layout == 0 || in_layout == 1);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU || act == MI355_ACT_RELU);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    MI355_CHECK_ARG(OH > 0 && OW > 0);
    const int Kreal = Cin * KH * KW, K = (Kreal + 3) & ~3;
    MI355_CHECK_ARG(ldw >= K && (ldw & 3) == 0);
    if (!aligned16(x) || !aligned16(weight) || (in_layout == 1 && (Cin & 3)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_conv2d_tokens_fwd: 16-byte aligned buffers and, for token-major input, Cin %% 4 == 0 (Cin=%d)", Cin);
    if (act != MI355_ACT_RELU && mi355::opt_stem_direct() && mi355::stem_conv_applicable(Cin, Cout, KH, KW, in_layout, bias, pos, y)) {
        const int rc = mi355::stem_conv(x, weight, bias, pos, y, B, Cin, H, W, Cout, KH, KW, stride, pad, ldw, in_layout, act,
                                        static_cast<hipStream_t>(stream));       // narrow stem layers: direct fp32 kernel (stem_conv.hip)
        if (rc) return rc;
        MI355_LAUNCH_CHECK();
        return MI355_OK;
    }
    GemmArgs g{};
    g.A = x; g.B = weight; g.C = y; g.bias = bias; g.pos = pos; g.act = act;
    g.P = OH * OW; g.Pout = g.P; g.OW = OW; g.Kreal = Kreal;
    g.M = B * g.P; g.N = Cout; g.K = K; g.lda = 0; g.ldb = ldw; g.ldc = Cout;
    g.Cin = Cin; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = in_layout == 0 ? launch<0, 2>(g, 1, precision, st) : launch<0, 3>(g, 1, precision, st);
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
