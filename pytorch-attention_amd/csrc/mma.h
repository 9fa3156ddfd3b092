// mma.h -- MFMA operand traits for the three precision modes (gfx950, wave64).
//
//   PREC 1 (fp16)   : operands rounded to IEEE half (RNE), v_mfma_f32_16x16x32_f16, fp32 accumulate
//   PREC 2 (bf16)   : operands rounded to bfloat16 (RNE), v_mfma_f32_16x16x32_bf16
//   PREC 0 (strict) : x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a.b ~ hi.hi + hi.lo + lo.hi (3 MFMAs),
//                     i.e. ~16 mantissa bits per operand -> fp32-class results at 1/3 of the bf16 rate
//
// 16x16x32 fragment maps (cdna_hip_programming.md section 3): A lane l holds row (l & 15), k = (l >> 4) * 8 + [0,8);
// B lane l holds column (l & 15), same k; C/D lane l holds column (l & 15), rows (l >> 4) * 4 + [0,4).
#pragma once
#include <hip/hip_runtime.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

template <int PREC>
struct Mma;

template <>
struct Mma<1> {
    static constexpr int NSPLIT = 1;
    using v8 = h8;
    using v4 = h4;
    using e = _Float16;
    __device__ static __forceinline__ v4 cvt(f4 v) { return v4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }
    __device__ static __forceinline__ e cvt1(float v) { return (_Float16)v; }
    __device__ static __forceinline__ f4 mma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <>
struct Mma<2> {
    static constexpr int NSPLIT = 1;
    using v8 = b8;
    using v4 = b4;
    using e = __bf16;
    __device__ static __forceinline__ v4 cvt(f4 v) { return v4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; }
    __device__ static __forceinline__ e cvt1(float v) { return (__bf16)v; }
    __device__ static __forceinline__ f4 mma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <>
struct Mma<0> : Mma<2> {
    static constexpr int NSPLIT = 2;
    __device__ static __forceinline__ v4 cvt_lo(f4 v, v4 hi) {
        return v4{(__bf16)(v.x - (float)hi.x), (__bf16)(v.y - (float)hi.y), (__bf16)(v.z - (float)hi.z),
                  (__bf16)(v.w - (float)hi.w)};
    }
    __device__ static __forceinline__ e cvt1_lo(float v, e hi) { return (__bf16)(v - (float)hi); }
};

// acc += A.B for one 16x16x32 step, in the operand format of PREC (hi/lo pairs for strict)
template <int PREC>
__device__ __forceinline__ f4 mma_step(const typename Mma<PREC>::v8* a, const typename Mma<PREC>::v8* b, f4 acc) {
    if constexpr (Mma<PREC>::NSPLIT == 2) {
        acc = Mma<PREC>::mma(a[1], b[0], acc);   // lo.hi
        acc = Mma<PREC>::mma(a[0], b[1], acc);   // hi.lo
    }
    return Mma<PREC>::mma(a[0], b[0], acc);       // hi.hi
}
