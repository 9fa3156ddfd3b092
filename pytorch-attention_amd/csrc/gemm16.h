// gemm16.h -- argument block and MFMA operand traits shared by the 16-bit GEMM kernels (gemm16.hip, gemm16_p8.hip).
#pragma once
#include "common.h"
#include "mma.h"

namespace g16 {

constexpr int BK = 64;

struct G16Args {
    const void* A; const void* B; void* C;
    const float* bias; const float* gamma; const float* resid;
    int M, N, K, lda, ldb, ldc;
    int act;
    int tr_rows;        // TR kernels only: rows per image (see the TR epilogue)
    int resid_period;   // gemm16_p8 fp32 outputs only, >= 128: the residual is a (resid_period, ldc) table read at row m % resid_period
                        // (position rows of a patch embedding); 0 = one residual row per output row
    unsigned* ovf;      // fp16 range guard word (common.h rg_report) or null; set by the entry points for 16-bit fp16 outputs
    const float* Af;    // LNA kernels only: fp32 activation rows (row stride lda floats), normalised on the way into LDS
    float ln_eps;
    // LayerNorm fold (ln_fold.hip), PRODUCER side -- gemm16_pa fp32 outputs only: beside Y the epilogue emits the next GEMM's operand
    //   lnc_a[m][n] = T(Y[m][n] - lnc_c[m])    (16-bit, row stride lnc_lda)   and per (32-column group g, row m) the pair
    //   lnc_stats[g * M + m] = {mean_g, sum_g (Y - mean_g)^2}                  (the exact row statistics, combined by mi355_ln_finalize_fwd)
    void* lnc_a; int lnc_lda; float* lnc_stats; const float* lnc_c;
    // ... CONSUMER side -- gemm16_p8 16-bit outputs only:  Y = act(rowtau[m].x * acc + rowtau[m].y * colsum[n] + bias[n])
    const float* rowtau; const float* colsum;
    // gemm16_wreg only (round 6): beside Y the kernel writes the LayerNorm statistics of every OUTPUT row, row_stats[2 m] = mean,
    // row_stats[2 m + 1] = 1 / sqrt(var + ln_eps) (two-pass, biased variance: the expression of ln_stats_kernel) -- the rows are complete
    // inside one workgroup there, so the statistics pass over Y that the next block would launch (XCABlock: norm3 in front of LPI) is free
    float* row_stats;
    // ... or, instead, the NEXT LayerNorm applied to the output rows and written in the operand format (16-bit, row stride ln16_ld):
    //   ln16_out[m][n] = T( (Y[m][n] - mean_m) * rstd_m * ln16_w[n] + ln16_b[n] )  -- the expression of layernorm_kernel; CSWin stage 3:
    // proj + residual and norm2 in one launch (cswin.py:192-194).  ovf (above) receives the fp16 range report of these conversions (code 2).
    void* ln16_out; const float* ln16_w; const float* ln16_b; int ln16_ld;
};

template <typename T> struct Vec8;
template <> struct Vec8<_Float16> { using t = h8; using t4 = h4; };
template <> struct Vec8<__bf16> { using t = b8; using t4 = b4; };

template <typename T>
__device__ __forceinline__ f4 mma16(typename Vec8<T>::t a, typename Vec8<T>::t b, f4 c);
template <>
__device__ __forceinline__ f4 mma16<_Float16>(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
template <>
__device__ __forceinline__ f4 mma16<__bf16>(b8 a, b8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

}  // namespace g16

namespace mi355 {
int gemm16_p8(const g16::G16Args& g, int out16, int precision, void* ws, size_t ws_bytes, hipStream_t st);     // gemm16_p8.hip
size_t gemm16_p8_workspace_bytes(int M, int N, int K);
int gemm16_pa(const g16::G16Args& g, int out16, int precision, hipStream_t st, int abl = 0);        // g.lnc_a != null: emitting variant                                 // gemm16_pa.hip
int gemm16_w4(const g16::G16Args& g, int out16, int precision, hipStream_t st);                        // gemm16_w4.hip
int gemm16_wreg(const g16::G16Args& g, int out16, int precision, hipStream_t st);                      // gemm16_wreg.hip
int gemm16_wst(const g16::G16Args& g, int out16, int precision, hipStream_t st);                       // gemm16_wst.hip
int gemm16_wslab(const g16::G16Args& g, int out16, int precision, hipStream_t st);                     // gemm16_wslab.hip
int gemm16_wslab_check(const g16::G16Args& g, int precision);                                            // would gemm16_wslab take it? (nothing launched)
int linear16_dispatch(const g16::G16Args& g, int out16, int precision, void* ws, size_t ws_bytes, hipStream_t st);    // gemm16.hip
}
