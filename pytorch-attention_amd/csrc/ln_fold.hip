// ln_fold.hip -- LayerNorm folded into the two GEMMs around it (ViT encoder chain, ViT.py:116-119: x + attn(LN1 x), x + mlp(LN2 x)).
//
// A pre-LN block spends one memory-bound launch per LayerNorm between two GEMMs (24 x 40 us of the 12 ms ViT-Base forward at
// B = 256: read the fp32 stream, write the 16-bit operand).  The fold removes the launch and the extra read:
//
//   producer  (the GEMM that writes the residual stream: proj / fc2, gemm16_pa.hip template flag LNC): beside Y = resid + act(..) its
//             epilogue emits the NEXT GEMM's operand  a[m][k] = T(Y[m][k] - c[m])  with c[m] = the row's mean BEFORE this update (known
//             exactly from the previous LayerNorm), and per (row, 32-column group) the exact pair (mean_g, sum (Y - mean_g)^2);
//   finalize  (this file, lane = row, ~10 MB of traffic): combines the 24 group pairs of a row (Chan's parallel update, fixed order)
//             into the exact mean mu and rstd r of the NEW row and writes rowtau[m] = {r, r (c - mu)} and c[m] := mu for the next producer;
//   consumer  (qkv / fc1, gemm16_p8.hip template flag FOLD) runs on `a` with the LayerNorm gain folded into its weights
//             W'[n][k] = T(gamma[k] W[n][k]) and finishes in its epilogue
//                 y[m][n] = r (sum_k (x - mu) W') + b' = rho[m] acc[m][n] + tau[m] colsum[n] + b'[n],   colsum[n] = sum_k W'[n][k],
//                 b'[n] = b[n] + sum_k beta[k] W[n][k]
//             -- a rank-1 correction, no extra K columns.
//
// Why c = the old mean and not 0: fp16(x) carries |mean| / std times the relative error of fp16(LayerNorm(x));
// fp16(x - c) carries (1 + |c - mu| / std), and the residual update of one sub-block rarely moves a row's mean by more than a
// fraction of its std.  "Rarely" is made "never" here: finalize checks |c - mu| <= tol * std and the fp16 range band of the row
// (std in [2^-7, 2^10]: nothing in x - c overflows, nothing relevant is subnormal); a row that fails is rewritten from the fp32
// stream as a = T((x - mu) r), rowtau = {1, 0} -- the plain LayerNorm operand -- by the whole wave.  So the result never depends on
// the heuristic, only the time does (measured on the ViT-Base forward: no row takes the slow path).
// The first LayerNorm of a chain has no producer: mi355_ln_center16_fwd writes the plain operand for every row.
#include <type_traits>
#include "common.h"
#include "gemm16.h"

namespace {
using v4f = float __attribute__((ext_vector_type(4)));
typedef float f2_ __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ void store4h(T* p, v4f v) {
    typedef T o4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<o4*>(p) = o4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
}

// plain operand of one row by one wave: a = T((x - mu) r); |a| <= sqrt(cols): no range concern
template <typename T>
__device__ __forceinline__ void write_row_normalised(const float* __restrict__ xr, T* __restrict__ ar, int cols, float mu, float r) {
    const int lane = threadIdx.x & 63;
    for (int i = lane * 4; i < cols; i += 256) {
        const v4f v = *reinterpret_cast<const v4f*>(xr + i);
        store4h<T>(ar + i, (v - mu) * r);
    }
}

// first LayerNorm of a chain: one wave per row, the row in registers (cols <= 2048, cols % 4 == 0), two-pass statistics like
// layernorm.hip; output = the plain operand, rowtau = {1, 0}, c = mean
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_center16_kernel(const float* __restrict__ x, T* __restrict__ a, float* __restrict__ rowtau,
                                                         float* __restrict__ cvec, long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int n4 = cols >> 2;
    const float inv = 1.0f / (float)cols;
    for (long row = wave0; row < rows; row += nwaves) {
        const v4f* xr = reinterpret_cast<const v4f*>(x + row * cols);
        v4f v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < n4 ? xr[i] : v4f{0.f, 0.f, 0.f, 0.f};
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (lane + 64 * j < n4) {
                const v4f d = v[j] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        T* ar = a + row * cols;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < n4) store4h<T>(ar + 4 * i, (v[j] - mean) * rstd);
        }
        if (lane == 0) {
            *reinterpret_cast<f2_*>(rowtau + 2 * row) = f2_{1.f, 0.f};
            cvec[row] = mean;
        }
    }
}

// finalize: EIGHT lanes per row (a wave = 8 rows): lane `sub` of a row combines groups sub, sub + 8, sub + 16, ..., the eight partial
// results meet through three DPP steps (fixed tree: the result never depends on the launch geometry).  With one lane per row the
// kernel was 197 workgroups of 48 dependent loads each: 16 us for 10 MB.
// stats[g * rows + m] = {mean_g, M2_g} over the 32 columns of group g (written by the producer's epilogue).
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ x, T* __restrict__ a,
                                                         float* __restrict__ cvec, float* __restrict__ rowtau, int rows, int cols, float eps,
                                                         float tol, unsigned* slow_rows) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool live = row < rows;
    const int G = cols >> 5;
    const f2_* st = reinterpret_cast<const f2_*>(stats);
    constexpr int MAXJ = 8;                                  // cols <= 2048: at most 64 groups = 8 per lane
    f2_ p[MAXJ];
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int g = sub + 8 * j;
        p[j] = (live && g < G) ? st[(long)g * rows + row] : f2_{0.f, 0.f};
        sm += p[j].x;
    }
    const float mu = sum8_dpp(sm) / (float)G;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        if (sub + 8 * j < G) {
            const float d = p[j].x - mu;
            m2 += p[j].y + 32.0f * (d * d);
        }
    }
    const float var = sum8_dpp(m2) / (float)cols;
    const float r = 1.0f / sqrtf(var + eps);
    bool bad = false;
    if (live && sub == 0) {
        const float sd = sqrtf(var), c = cvec[row];
        // every comparison is written so that a NaN anywhere selects the slow path (whose output is then the reference's NaN row)
        bad = !(fabsf(c - mu) <= tol * sd) || !(sd <= 1024.0f) || !(sd >= 0.0078125f);
        *reinterpret_cast<f2_*>(rowtau + 2 * (long)row) = bad ? f2_{1.f, 0.f} : f2_{r, r * (c - mu)};
        cvec[row] = mu;
    }
    // rows outside the band: the plain LayerNorm operand from the fp32 stream, one row at a time by the whole wave
    unsigned long long mask = __ballot(bad);
    if (mask && lane == 0 && slow_rows) atomicAdd(slow_rows, (unsigned)__popcll(mask));
    while (mask) {
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int rr = __builtin_amdgcn_readlane(row, l);
        const float rmu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu), l));
        const float rrs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), l));
        write_row_normalised<T>(x + (long)rr * cols, a + (long)rr * cols, cols, rmu, rrs);
    }
}

}  // namespace

extern "C" {

size_t mi355_ln_fold_stats_bytes(int rows, int cols) { return rows > 0 && cols > 0 ? (size_t)(cols / 32) * rows * 8 : 0; }

int mi355_ln_center16_fwd(const float* x, void* a16, float* rowtau, float* cvec, int rows, int cols, float eps, int precision,
                          mi355_stream_t stream) {
    MI355_CHECK_ARG(x && a16 && rowtau && cvec && rows > 0 && cols > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((cols & 3) || cols > 2048 || !aligned16(x) || !aligned16(a16) || (reinterpret_cast<uintptr_t>(rowtau) & 7u))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_ln_center16_fwd: needs cols %% 4 == 0, cols <= 2048, 16-byte aligned rows (cols=%d)", cols);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (int)(((long)rows + 3) / 4 < 8192 ? ((long)rows + 3) / 4 : 8192);
    MI355_TRACE(st, "ln_center16_kernel rows=%d cols=%d", rows, cols);
#define LC(T_, NV_) ln_center16_kernel<T_, NV_><<<grid, 256, 0, st>>>(x, static_cast<T_*>(a16), rowtau, cvec, (long)rows, cols, eps)
#define LC_BY_NV(T_) do { if (cols <= 256) LC(T_, 1); else if (cols <= 512) LC(T_, 2); else if (cols <= 1024) LC(T_, 4); else LC(T_, 8); } while (0)
    if (precision == MI355_PREC_FP16) LC_BY_NV(_Float16);
    else LC_BY_NV(__bf16);
#undef LC_BY_NV
#undef LC
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_ln_finalize_fwd(const float* stats, const float* x, void* a16, float* cvec, float* rowtau, int rows, int cols, float eps,
                          float tol, int precision, unsigned* slow_rows, mi355_stream_t stream) {
    MI355_CHECK_ARG(stats && x && a16 && cvec && rowtau && rows > 0 && cols > 0 && tol >= 0.f);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((cols & 31) || !aligned16(x) || !aligned16(a16) || (reinterpret_cast<uintptr_t>(rowtau) & 7u) || (reinterpret_cast<uintptr_t>(stats) & 7u))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_ln_finalize_fwd: needs cols %% 32 == 0 and aligned buffers (cols=%d)", cols);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (rows + 31) / 32;                          // 4 waves x 8 rows per workgroup
    MI355_TRACE(st, "ln_finalize_kernel rows=%d cols=%d", rows, cols);
    if (precision == MI355_PREC_FP16)
        ln_finalize_kernel<_Float16><<<grid, 256, 0, st>>>(stats, x, static_cast<_Float16*>(a16), cvec, rowtau, rows, cols, eps, tol, slow_rows);
    else
        ln_finalize_kernel<__bf16><<<grid, 256, 0, st>>>(stats, x, static_cast<__bf16*>(a16), cvec, rowtau, rows, cols, eps, tol, slow_rows);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// producer: Y = resid + act(X16 W16^T + bias) on the two-accumulator persistent kernel, emitting the next operand and the group statistics
int mi355_linear16_emit_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K, int ldx,
                            int act, int precision, const float* cvec, void* a16_out, float* stats, mi355_stream_t stream) {
    MI355_CHECK_ARG(X16 && W16 && Y && cvec && a16_out && stats && M > 0 && N > 0 && K > 0 && ldx >= K);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((ldx & 7) || !aligned16(X16) || !aligned16(W16) || !aligned16(Y) || !aligned16(a16_out) || (bias && !aligned16(bias)) ||
        (resid && !aligned16(resid)) || (reinterpret_cast<uintptr_t>(stats) & 7u))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_emit_fwd: 16-byte aligned rows required");
    g16::G16Args g{};
    g.A = X16; g.B = W16; g.C = Y; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = N; g.act = act;
    g.lnc_a = a16_out; g.lnc_lda = N; g.lnc_stats = stats; g.lnc_c = cvec;
    const int rc = mi355::gemm16_pa(g, 0, precision, static_cast<hipStream_t>(stream));
    if (rc == MI355_EUNSUPPORTED)
        return mi355::fail(rc, "mi355_linear16_emit_fwd: needs M %% 128 == 0, N %% 256 == 0, K %% 64 == 0, K >= 640 (M=%d N=%d K=%d)", M, N, K);
    if (rc != MI355_OK) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// consumer: Y16 = act(rowtau.x * (A16 W16'^T) + rowtau.y * colsum + bias) on the persistent 256 x 256 kernel
int mi355_linear16_lnfold_fwd(const void* A16, const void* W16, const float* bias, const float* rowtau, const float* colsum, void* Y16,
                              int M, int N, int K, int lda, int ldy, int act, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(A16 && W16 && rowtau && colsum && Y16 && M > 0 && N > 0 && K > 0 && lda >= K && ldy >= N);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((lda & 7) || (ldy & 7) || !aligned16(A16) || !aligned16(W16) || !aligned16(Y16) || (bias && !aligned16(bias)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_lnfold_fwd: 16-byte aligned rows required");
    g16::G16Args g{};
    g.A = A16; g.B = W16; g.C = Y16; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldy; g.act = act;
    g.rowtau = rowtau; g.colsum = colsum;
    if (precision == MI355_PREC_FP16) g.ovf = mi355::range_word(static_cast<hipStream_t>(stream));
    const int rc = mi355::gemm16_p8(g, 1, precision, nullptr, 0, static_cast<hipStream_t>(stream));
    if (rc == MI355_EUNSUPPORTED) return mi355::fail(rc, "mi355_linear16_lnfold_fwd: needs K %% 64 == 0, N %% 8 == 0 (N=%d K=%d)", N, K);
    if (rc != MI355_OK) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
