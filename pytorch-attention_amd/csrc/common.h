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
long  opt_gemm_variant();
long  opt_gemm_splitk();
long  opt_eca_single();
long  opt_se_single();
long  opt_se_occ();
long  opt_ws_persistent();
struct WsEpoch { unsigned tag; unsigned ticket_base; bool fresh; };
WsEpoch ws_epoch(const void* region, unsigned long long key, unsigned draws, hipStream_t st);   // api.hip: tag + ticket base of this launch
void  ws_forget(const void* region);
bool  stream_is_capturing(hipStream_t st);      // hipGraph capture in progress on this stream
void  ws_forget_range(const void* base, size_t bytes);
long  opt_cbam_single();
// ---- exchange-kernel failure reporting (api.hip) ----------------------------------------------------------------------------
// The single-read kernels bound their inter-workgroup polls.  A poll that runs out stores a non-zero code into ONE pinned,
// device-visible host word (system-scope store, no synchronisation needed to read it); every later library entry that launches an
// exchange kernel -- and mi355_sync_status() -- looks at that word first and fails with MI355_ESYNC instead of returning OK over
// garbage.  spin_limit() is the poll budget (option "spin_limit", default 1 << 22 sweeps ~ a second).
unsigned* sync_err_word();            // device-visible pinned host word (null if the allocation failed: reporting falls back to the workspace word)
unsigned  spin_limit();
int   sync_pending(const char* who);  // MI355_OK, or MI355_ESYNC with the error text set (the word is cleared: reported once)
int   resident_slots(int per_cu);     // multiprocessor count of the current device x per_cu
long  opt_zoo_single();
long  opt_stem_direct();
bool  stem_conv_applicable(int Cin, int Cout, int KH, int KW, int in_layout, const float* bias, const float* pos, const float* y);
int   stem_conv(const float* x, const float* w, const float* bias, const float* pos, float* y, int B, int Cin, int H, int W, int Cout, int KH,
                int KW, int stride, int pad, int ldw, int in_layout, int act, hipStream_t st);
size_t zoo_workspace_bytes(int B, int C);
size_t cbam_single_extra_bytes(int B, int C, int H, int W);
bool  cbam_single_applicable(int C, int Cr, int H, int W, int ks);
int   cbam_single(const float* x, const float* w1, const float* w2, const float* wconv, float* y, int B, int C, int Cr, int H, int W,
                  int ks, void* extra, hipStream_t st);
size_t se_single_extra_bytes(int B, int C);
bool  se_single_applicable(int C, int Cr, int H, int W);
// SE variants embedded in the reference's CNNs: optional biases of the two excitation layers and the gate function
struct SeExtra {
    const float* b1;     // (Cr) or null
    const float* b2;     // (C) or null
    int gate;            // 0 sigmoid, 1 hard sigmoid relu6(z + 3) / 6
};
int   se_single(const float* x, const float* w1, const float* w2, float* y, int B, int C, int Cr, int H, int W, void* state,
                void* gran, SeExtra ex, hipStream_t st);
bool  eca_single_applicable(int C, int k, int H, int W);
int   eca_single(const float* x, const float* taps, float* y, int B, int C, int k, int H, int W, hipStream_t st);
size_t fused_state_bytes(int B);
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
// Logical workgroup id of a 1-D grid such that each XCD works on one CONTIGUOUS range of logical ids (the hardware deals consecutive
// blockIdx.x to the 8 XCDs round-robin): neighbours in the logical order -- heads of one image, query blocks of one head -- then share
// an L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_contiguous_block() {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
// ReLU with torch's NaN behaviour (relu(NaN) = NaN, so one NaN in x poisons the whole gate exactly as in the reference): v_max_f32
// returns the non-NaN operand, and a float compare could be folded away under -fno-honor-nans, hence the bit test.
__device__ __forceinline__ float relu_nan(float v) {
    const float r = fmaxf(v, 0.f);
    return ((__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u) ? v : r;
}
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float se_gate(float z, int kind) {
    if (!kind) return sigmoidf_(z);
    const float h = fminf(fmaxf(z + 3.0f, 0.0f), 6.0f) / 6.0f;                    // relu6 clamps; torch's clamp keeps a NaN
    return ((__float_as_uint(z) & 0x7fffffffu) > 0x7f800000u) ? z : h;
}
// exact-erf GELU (nn.GELU() default), library erff: used off the hot path (LPI)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-form GELU for the GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level) on the
// v_rcp_f32 / v_exp_f32 units -- ~15 instructions instead of ~40 for erff, which matters because the epilogue is not
// overlapped with MFMA work (profiles: the erff epilogue cost more than the 768-deep MFMA loop of the fc1 GEMM).
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    return 0.5f * x * (1.0f + __builtin_copysignf(erf_abs, z));
}
#endif
