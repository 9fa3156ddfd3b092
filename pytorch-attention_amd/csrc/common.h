// common.h -- shared host/device helpers for libmi355attn (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/mi355attn.h"

#define MI355_ABI_VERSION 1
#define WAVE 64

// ---- error plumbing (host) --------------------------------------------------------------------------
namespace mi355 {
char* err_buf();                      // thread-local 512-byte buffer (defined in api.hip)
int   fail(int code, const char* fmt, ...);
long  opt_chunk_images();
long  opt_nt();
long  opt_reverse();
long  opt_fused();
size_t fused_state_bytes(int B);
bool  fused_applicable(int B, int C, int H, int W);
int   se_eca_fused(int mode, const float* x, const float* wa, const float* wb, float* y, int B, int C, int Cr, int H, int W,
                   float* means, void* state, hipStream_t st);
// GEMM engine (gemm.hip), shared by the other translation units.  NT: B is (N,K) K-contiguous; KN: B is (K,N) N-contiguous.
int gemm_nt(const float* A, const float* B, const float* bias, const float* gamma, const float* resid, float* C, int M, int N,
            int K, int lda, int ldb, int ldc, int act, int precision, hipStream_t st);
int gemm_nt_batched(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc, long sA,
                    long sB, long sC, int precision, hipStream_t st);
int gemm_kn_batched(const float* A, const float* B, const float* bias_row, const float* resid, float* C, int batch, int M,
                    int N, int K, int lda, int ldb, int ldc, long sA, long sB, long sC, int act, int precision,
                    hipStream_t st);
}  // namespace mi355

#define MI355_CHECK_ARG(cond)                                                                   \
    do {                                                                                        \
        if (!(cond)) return mi355::fail(MI355_EINVAL, "%s: invalid argument: %s", __func__, #cond); \
    } while (0)

#define MI355_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return mi355::fail(MI355_EHIP, "%s: %s -> %s", __func__, #call, hipGetErrorString(e_)); \
    } while (0)

#define MI355_LAUNCH_CHECK()                                                                    \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess)                                                                   \
            return mi355::fail(MI355_EHIP, "%s: kernel launch -> %s", __func__, hipGetErrorString(e_)); \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int  cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
// exact-erf GELU (nn.GELU() default)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
#endif
