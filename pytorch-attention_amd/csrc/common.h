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
long  opt_gemm_pa();
long  opt_da_fused();
long  opt_da_ranges();
long  opt_eca_single();
long  opt_se_single();
long  opt_se_occ();
long  opt_ws_persistent();
long  opt_ln_fold();
long  opt_gemm_pa16();
long  opt_gemm_pa_block();
long  opt_gemm_pa_tail();
long  opt_lpi_patch();
long  opt_mixer_early();
long  opt_gemm_small();
long  opt_mlp_tt4();
long  opt_mixer_stats();
long  opt_attn_nw();
long  opt_gemm_w4();
long  opt_gemm_wreg();
long  opt_xca_tr();
long  opt_mlp_wide();
long  opt_gemm_wst();
long  opt_gemm_wslab();
// fused LayerNorm + MLP for C = 256 / 384 (mlp_wide.hip): waves split the weights, fragments go global -> VGPR
bool  mlp_wide_applicable(int C, int hidden);
int   mlp_wide(const float* x, const void* w1_16, const float* b1, const void* w2_16, const float* b2, const float* gamma, float* y, long M, int C,
               int layernorm, float eps, int precision, hipStream_t st);
// one-wave 16 x 32 tiles for small outputs (gemm_small.hip): MI355_EUNSUPPORTED when the shape is the engine's
int   gemm_small_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc, int precision,
                    hipStream_t st);
void  ws_forget(const void* region);
bool  ws_known(const void* region, unsigned long long key, hipStream_t st);   // api.hip: was this workspace zeroed for this shape? (device-resident launch tags)
hipError_t ws_zero_async(void* p, size_t bytes, hipStream_t st);   // zero an exchange area with a kernel (capture-safe ordering)
unsigned* sync_err_word_on(hipStream_t st);     // sync_err_word(), but never allocates inside a stream capture (may return null there)
bool  stream_is_capturing(hipStream_t st);      // hipGraph capture in progress on this stream
void  ws_forget_range(const void* base, size_t bytes);
long  opt_cbam_single();
// ---- exchange-kernel failure reporting (api.hip) ----------------------------------------------------------------------------
// The single-read kernels bound their inter-workgroup polls.  A poll that runs out stores a non-zero code into ONE pinned,
// device-visible host word (system-scope store, no synchronisation needed to read it); every later library entry that launches an
// exchange kernel -- and mi355_sync_status() -- looks at that word first and fails with MI355_ESYNC instead of returning OK over
// garbage.  spin_limit() is the poll budget (option "spin_limit", default 1 << 22 sweeps ~ a second).
unsigned* sync_err_word();            // device-visible pinned host word (null if the allocation failed: reporting falls back to the workspace word)
// fp16 range guard (round 3): the producers of fp16 OPERAND tensors (mi355_cast16_fwd, mi355_layernorm16_fwd, the 16-bit-output GEMM
// epilogues) keep the largest magnitude they convert; a finite value that saturates to inf in fp16 (|v| >= 65520) stores a code into
// a second pinned host word.  mi355_range_status() reads it WITHOUT a device synchronisation (like mi355_sync_status); the Python
// binding raises before the next 16-bit launch.  bf16 operands have the fp32 range and are not checked.  Null under hipGraph
// capture when the word does not exist yet (hipHostMalloc is illegal there): the launch then runs unguarded.
unsigned* range_word(hipStream_t st);
int   range_pending(const char* who);  // MI355_OK or MI355_ERANGE (reported once)
// round 6 (api.hip): while the device is armed (mi355_range_arm), launchers that call range_word() are counted; ONE event is recorded in front
// of the first instrumented launch behind the predicted last producer (or at the tail); mi355_range_wait() waits for it and returns the status
void  range_mark_before_launch();
void  range_mark_entry_done();
int   range_arm(int on);
int   range_wait();
long  range_launches();
unsigned  spin_limit();
int   sync_pending(const char* who);  // MI355_OK, or MI355_ESYNC with the error text set (the word is cleared: reported once)
int   resident_slots(int per_cu);     // multiprocessor count of the current device x per_cu
// ---- in-process kernel tally (api.hip; mi355_trace_begin / mi355_trace_end) ---------------------------------------------------
// While a trace is open on the calling thread's device, a TraceScope brackets ONE kernel launch with a pair of HIP events on the
// launch stream and files the interval under a tag (kernel name + the shape parameters that tell its launches apart).  Closed: one
// relaxed atomic load per launch.  Never active under hipGraph stream capture.
bool  trace_on();
int   trace_begin();
long  trace_end(char* buf, size_t n);
struct TraceScope {
    int idx;
    hipStream_t st;
    TraceScope(hipStream_t st_, const char* fmt, ...);
    ~TraceScope();
    TraceScope(const TraceScope&) = delete;
    TraceScope& operator=(const TraceScope&) = delete;
};
#define MI355_TRACE(st, ...) mi355::TraceScope trace_scope_(st, __VA_ARGS__)
int   func_dynamic_lds(const void* fn, int bytes);   // hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): api.hip
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
// DoubleAttention in two passes over the image (double_attn_fused.hip): 16-bit operand modes, c_m = c_n = 128, C in {128, 256}
bool   double_attn_fused_ok(int B, int C, int cm, int cn, int HW, int precision);
size_t double_attn_fused_workspace(int B, int C, int HW);
int    double_attn_fused(const float* x, const float* wA, const float* bA, const float* wB, const float* bB, const float* wV, const float* bV,
                         const float* wP, const float* bP, float* y, int B, int C, int HW, int precision, void* ws, hipStream_t st);
// DoubleAttention in one kernel for c_m = c_n = 32, C = 64, H*W <= 1024 (double_attn_small.hip): the README / smoke-test shape class
bool   double_attn_small_ok(int B, int C, int cm, int cn, int HW, int precision);
int    double_attn_small(const float* x, const float* wA, const float* bA, const float* wB, const float* bB, const float* wV, const float* bV,
                         const float* wP, const float* bP, float* y, int B, int C, int HW, int precision, hipStream_t st);
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
        mi355::range_mark_entry_done();                                                         \
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
// wave_sum() of common.h without address registers: the xor-32 step is one v_permlane32_swap_b32 on two copies of the value (row 0 of
// one meets row 1 of the other), the xor-16 .. xor-1 steps are ds_swizzle_b32 in bit-mask mode (pattern in the instruction).  Same
// pairs in the same order as the __shfl_xor butterfly, fp32 addition is commutative: bit-identical sums.  The six ds_bpermute address
// registers of the __shfl_xor form were loop invariants the 80-register SE kernel had to spill a row chunk for.
template <int XOR> __device__ __forceinline__ float swz_xor(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1f));
}
__device__ __forceinline__ float wave_sum_sw(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    v = a + b;
    v += swz_xor<16>(v);
    v += swz_xor<8>(v);
    v += swz_xor<4>(v);
    v += swz_xor<2>(v);
    v += swz_xor<1>(v);
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
// ---- fp16 range guard (device side): running |max| of the values a lane converts to fp16; v_max ignores NaN operands --------------
typedef float rg_f4 __attribute__((ext_vector_type(4)));
// A group of four whose largest magnitude is inf is skipped instead of poisoning the lane's running maximum: the inf is the INPUT's
// (not reported: the reference holds it too), and a later finite value that saturates must still be seen.  (An inf and a saturating
// finite value inside the same group of four are not told apart.)
// The running maximum is carried as the BIT PATTERN of the magnitude plus 2^23 in a float-typed register and compared as a signed integer:
// finite magnitudes map monotonically into [2^23, 2^31), inf (0x7f800000) and NaN wrap to negative numbers and lose every integer
// maximum -- the skip costs one integer add instead of a compare + select per group (round 4's form made every 16-bit GEMM epilogue
// 7 % longer in instructions; the exposed epilogue of the persistent 256 x 256 kernel paid it in full: qkv of ViT-Base +2-6 %).
__device__ __forceinline__ float rg_fold(float m, float g) {
    const int u = (int)(__float_as_uint(g) + 0x00800000u);
    const int mi = (int)__float_as_uint(m);
    return __uint_as_float((unsigned)(u > mi ? u : mi));
}
__device__ __forceinline__ float rg_absmax4(float m, rg_f4 v) {
    return rg_fold(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
__device__ __forceinline__ float rg_absmax1(float m, float v) { return rg_fold(m, fabsf(v)); }
// 65520 = the smallest magnitude that rounds to inf in IEEE half (the running maximum only ever holds finite values)
__device__ __forceinline__ void rg_report(float m, unsigned* word, unsigned code) {
    if (word && (int)__float_as_uint(m) >= (int)(0x477FF000u + 0x00800000u))         // bits(65520.0f) + 2^23
        __hip_atomic_store(word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Float form of the same guard (compare + select per group; the accumulator holds the magnitude itself).  gemm16_pa keeps it: its
// epilogue pieces ride in the next tile's main loop, where the two extra VALU instructions are free, and with the integer form the
// three-piece instantiations (K = 256) spilled 76 B per lane at their 256-register budget.
__device__ __forceinline__ float rg_absmax4_f(float m, rg_f4 v) {
    const float g = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    return g < __builtin_inff() ? fmaxf(m, g) : m;
}
__device__ __forceinline__ void rg_report_f(float m, unsigned* word, unsigned code) {
    if (word && m >= 65520.0f) __hip_atomic_store(word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Cheapest form for the VALU-bound fused block kernels (round 6: mlp_fused.hip, cswin_fused.hip): ONE v_max3_f32 with |.| source modifiers per
// two converted values, the running maximum held as a plain float (starts at 0).  v_max3 returns the non-NaN operands; an inf operand -- the
// input's own or a saturated product -- makes the maximum inf and IS reported (a false positive for an input that already holds inf costs
// a strict re-run whose result is the reference's, inf included).  Report with rg_report_f.
__device__ __forceinline__ float rg_max3abs(float m, float a, float b) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float rg_max3abs4(float m, rg_f4 v) { return rg_max3abs(rg_max3abs(m, v.x, v.y), v.z, v.w); }
__device__ __forceinline__ float se_gate(float z, int kind) {
    if (!kind) return sigmoidf_(z);
    const float h = fminf(fmaxf(z + 3.0f, 0.0f), 6.0f) / 6.0f;                    // relu6 clamps; torch's clamp keeps a NaN
    return ((__float_as_uint(z) & 0x7fffffffu) > 0x7f800000u) ? z : h;
}
// exact-erf GELU (nn.GELU() default), library erff: used off the hot path (LPI)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-form GELU for the epilogues (nn.GELU() default: 0.5 x (1 + erf(x / sqrt 2))), built for the VALU budget of a kernel whose
// epilogue is not hidden behind MFMA work:
//     erfc(t) = 2^q(t) on t in [0, 4] with q a degree-8 polynomial (weighted minimax fit of log2 erfc; fp32 Horner: |erf error|
//     <= 1.2e-7, i.e. rounding level; beyond t = 4 erfc < 2^-26)  =>  gelu(x) = 0.5 ((x + |x|) - a 2^{p(a)}),  a = min(|x|, 4 sqrt 2),
// with p(a) = q(a / sqrt 2) (the scale is folded into the coefficients).  One transcendental (v_exp_f32) instead of two (the
// Abramowitz-Stegun 7.1.26 form used before also needed v_rcp_f32) and a smaller error.  x + |x| (not max(x, 0)) keeps a NaN in x
// alive like torch.  Measured on the fc1 epilogue of ViT-Base (155 M evaluations, same box, same process): this form 0.371-0.374 ms,
// the 7.1.26 form 0.374-0.378, this form on pairs of elements with v_pk_fma_f32 (gelu_fast2) 0.357-0.381 -- the epilogue is bound by
// the instruction count (14-15 either way; the transcendental unit overlaps).  Round 4 measured the packed form properly
// (tools/valu_probe.hip: v_pk_fma_f32 delivers 1.64x the results of v_fma_f32 per unit time in a pure FMA loop -- 115 vs 70 TFLOP/s --
// so it is NOT half rate, as this comment used to say) and rebuilt gelu16_fast on pairs (the Horner chain is five v_pk_fma_f32 instead
// of five scalar v_fmaak_f32, which have no packed form because of their literal): the fc1 epilogue of ViT-Base and the fused CSWin
// MLPs did not move (88 us of GELU in a 371 us fc1; 185 / 174 us) -- the evaluation is bound by v_exp_f32 (quarter rate, a third of
// its ~50 cycles per value), v_min, the conversions and the slab round trip, not by FMA issue.  The scalar form stays.
// max |gelu error| 5e-7 over [-12, 12] (the rounding of the result itself); tools/fit_gelu.py reproduces fit and check.
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
#define MI355_GELU_C0 2.1717760034789535e-08f
#define MI355_GELU_C1 -1.1511057615280151f
#define MI355_GELU_C2 -0.45920491218566895f
#define MI355_GELU_C3 -0.05250502750277519f
#define MI355_GELU_C4 0.007075471803545952f
#define MI355_GELU_C5 -0.00014587005716748536f
#define MI355_GELU_C6 -0.00018254770839121193f
#define MI355_GELU_C7 3.862079029204324e-05f
#define MI355_GELU_C8 -2.7720548132492695e-06f
#define MI355_GELU_CLAMP 5.65685424949238f
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float a = fminf(ax, MI355_GELU_CLAMP);
    float p = MI355_GELU_C8;
    p = __builtin_fmaf(p, a, MI355_GELU_C7);
    p = __builtin_fmaf(p, a, MI355_GELU_C6);
    p = __builtin_fmaf(p, a, MI355_GELU_C5);
    p = __builtin_fmaf(p, a, MI355_GELU_C4);
    p = __builtin_fmaf(p, a, MI355_GELU_C3);
    p = __builtin_fmaf(p, a, MI355_GELU_C2);
    p = __builtin_fmaf(p, a, MI355_GELU_C1);
    p = __builtin_fmaf(p, a, MI355_GELU_C0);
    const float e = __builtin_amdgcn_exp2f(p);
    return 0.5f * ((x + ax) - a * e);
}
__device__ __forceinline__ gelu_f2 gelu_fast2(gelu_f2 x) {
    const gelu_f2 ax = __builtin_elementwise_abs(x);
    const gelu_f2 a = __builtin_elementwise_min(ax, gelu_f2{MI355_GELU_CLAMP, MI355_GELU_CLAMP});
    gelu_f2 p = gelu_f2{MI355_GELU_C8, MI355_GELU_C8};
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C7, MI355_GELU_C7});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C6, MI355_GELU_C6});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C5, MI355_GELU_C5});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C4, MI355_GELU_C4});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C3, MI355_GELU_C3});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C2, MI355_GELU_C2});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C1, MI355_GELU_C1});
    p = __builtin_elementwise_fma(p, a, gelu_f2{MI355_GELU_C0, MI355_GELU_C0});
    const gelu_f2 e = gelu_f2{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    return ((x + ax) - a * e) * 0.5f;
}
typedef float gelu_f4 __attribute__((ext_vector_type(4)));      // the same type as f4 of mma.h
__device__ __forceinline__ gelu_f4 gelu_fast4(gelu_f4 v) {
    return gelu_f4{gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w)};
}
// The same GELU for results that are rounded to 16 bits right away (hidden units of the fused MLPs, 16-bit fc1 outputs): the
// exponent polynomial has degree 5 and carries the factor 1/2 in its constant term (erfc(a / sqrt 2) / 2 = 2^{h(a)}), so
//     gelu(x) = (x + |x|) / 2 - a 2^{h(a)}:   min, 5 FMAs, v_exp_f32, one multiply and two FMAs (|x| and -a are source modifiers)
// = 10 issue slots + the transcendental against 15 for gelu_fast.  Absolute error <= 9e-7 over [-12, 12] (the fp32 rounding of the
// result on the positive side, 5e-7 of approximation error on the negative side): less than half an fp16 ulp wherever
// |gelu| >= 2e-3 and far below a bf16 ulp everywhere.  A NaN in x survives through the (x + |x|) / 2 term.  tools/fit_gelu.py.
#define MI355_GELU16_H0 -1.000037670135498f
#define MI355_GELU16_H1 -1.1507878303527832f
#define MI355_GELU16_H2 -0.4599926173686981f
#define MI355_GELU16_H3 -0.05182718485593796f
#define MI355_GELU16_H4 0.007084468379616737f
#define MI355_GELU16_H5 -0.0004732950183097273f
__device__ __forceinline__ float gelu16_fast(float x) {
    const float ax = fabsf(x);
    const float a = fminf(ax, MI355_GELU_CLAMP);
    float p = MI355_GELU16_H5;
    p = __builtin_fmaf(p, a, MI355_GELU16_H4);
    p = __builtin_fmaf(p, a, MI355_GELU16_H3);
    p = __builtin_fmaf(p, a, MI355_GELU16_H2);
    p = __builtin_fmaf(p, a, MI355_GELU16_H1);
    p = __builtin_fmaf(p, a, MI355_GELU16_H0);
    const float e = __builtin_amdgcn_exp2f(p);
    const float t = __builtin_fmaf(0.5f, ax, 0.5f * x);
    return __builtin_fmaf(-a, e, t);
}
__device__ __forceinline__ gelu_f4 gelu16_fast4(gelu_f4 v) {
    return gelu_f4{gelu16_fast(v.x), gelu16_fast(v.y), gelu16_fast(v.z), gelu16_fast(v.w)};
}
// 16-bit destination -> gelu16_fast, fp32 destination -> gelu_fast
template <bool OUT16>
__device__ __forceinline__ gelu_f4 gelu_out4(gelu_f4 v) {
    if constexpr (OUT16) return gelu16_fast4(v);
    else return gelu_fast4(v);
}
#endif
