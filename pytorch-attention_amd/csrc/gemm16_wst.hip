// gemm16_wst.hip -- Y16 (M x N, 16 bit) = act(X16 (M x 768) . W16^T (N x 768) + bias) with the WEIGHTS STATIONARY in registers: the qkv and fc1
// products of a ViT-Base encoder layer (ViT.py:81 and :59-61; N = 2304 / 3072, K = 768, M = 50 432 at B = 256), gfx950.  Round 6, OPT-IN.
//
// The tile kernels of the engine (gemm16_p8 / gemm16_w4 / gemm16_pa) stream BOTH operands through LDS for every 256 x 256 output tile:
// 1.39 GB cross the L2 -> CU path for the qkv product, half of it the same 3.5 MB of weights over and over, and the MFMA pipes sit 0.44-0.53
// busy (profiles/r06_c5_mfma_util.txt).  With K = 768 a 192-column slab of W is 295 KB -- it fits the register file of a CU:
//
//   workgroup  = 8 waves, persistent, owns ONE 192-column slab for the whole kernel and walks 32-row tiles of X (the workgroups of a slab
//                take the row tiles round-robin; the slabs of one row tile sit on one XCD, so X crosses HBM -> L2 once per XCD);
//   wave (p, h) p = wave >> 1: columns [48 p, 48 p + 48) of the slab; h = wave & 1: the K half [384 h, 384 h + 384).  Its 48 x 384 block of W
//                lives in VGPRs as MFMA A-fragments (3 column tiles x 12 k-steps x 4 = 144 registers), loaded once;
//   X tile       32 rows x 768 = 48 KB, LDS-DMA'ed as one linear block (16-byte chunks XOR-swizzled at the SOURCE: chunk c of row r sits at
//                chunk c ^ (r & 15), so the 16 rows a fragment read touches fall into 16 different bank groups), double-buffered; an X fragment
//                read from LDS feeds three MFMAs;
//   per tile     72 MFMAs per wave; the two K halves of a column block are added through LDS (wave h finishes row tile h: bias, GELU, 16 bit);
//                the 16-bit results wait in six registers and are stored one tile later, so that the one counted wait of an iteration never
//                covers a freshly issued store; two raw barriers.
// L2 -> CU bytes: X once per slab (12 x 77 MB for qkv) + W once per workgroup, against 1.39 GB for the 256 x 256 tiling.
// A row's K steps are added as (k < 384) + (k >= 384), each half in ascending order: NOT the bit pattern of the tile kernels (one chain), but
// the same for a row wherever it sits in the batch.
// MEASURED (profiles/r06_gemm_wst.md): qkv 208-219 us against 174-180 us on gemm16_w4, fc1 306-383 against 275-288 on gemm16_pa; a tie only on
// row counts the one-wave-per-SIMD kernel does not take (M % 256 != 0: 192-200 vs 192-204 us).  Two barriers, the K-half exchange and the
// per-tile DMA address arithmetic leave the matrix pipes 0.43 busy -- below the tile kernels' 0.53.  Option "gemm_wst" (default 0).
#include "gemm16.h"
#include "bufops.h"
#include <type_traits>

namespace {

using namespace g16;

__device__ __forceinline__ unsigned wst_lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}
// One 1 KB LDS-DMA (16 bytes per lane, lane-linear at `dst`), issued behind the compiler's back like double_attn_fused.hip's: through the
// builtin hipcc would drain the prefetch with `s_waitcnt vmcnt(0)` in front of unrelated LDS accesses; the kernel counts its DMAs itself.
__device__ __forceinline__ void wst_dma16(const void* src, const void* dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wst_lds_addr(dst)))
                 : "memory");
}
#define WST_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// 192-column slabs, 32-row tiles: 144 registers of W per wave leave room for two row tiles of accumulators and fragment reads three ahead.
// (Measured and removed: 256-column slabs with 16-row tiles -- 192 registers of W, twice the barriers per row: 244-262 us for the qkv product -- and
// one wave per SIMD with 64 columns x the whole K in 384 registers, no K split: bit-identical to the tile kernels, but hipcc keeps MFMA operands in
// VGPRs and shuttles the fragments through AGPRs with v_accvgpr moves: 545 us.  profiles/r06_gemm_wst.md)
template <typename T, bool GELU>
__global__ __launch_bounds__(512, 1) void gemm16_wst_kernel(const G16Args g, int nslab) {
    using v8 = typename Vec8<T>::t;
    typedef T t4 __attribute__((ext_vector_type(4)));
    constexpr int K = 768, KSH = 12, RT = 2, ROWS = 32, CPR = K / 8, NCT = 3, SLAB = 4 * NCT * 16;
    constexpr int TILE_EL = ROWS * K;                                 // 48 KB
    constexpr int NDMA = TILE_EL * 2 / 1024 / 8;                      // 6 per wave and tile
    constexpr int NB = 2;
    __shared__ __attribute__((aligned(1024))) unsigned short s_x[NB][TILE_EL];
    __shared__ __attribute__((aligned(16))) float s_ex[8][NCT][64 * 4];
    __shared__ __attribute__((aligned(16))) float s_bias[SLAB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, gq = lane >> 4;
    const int p = wave >> 1, h = wave & 1;
    const int lid = xcd_contiguous_block();
    const int slab = lid % nslab, stream = lid / nslab;
    const int nstream = ((int)gridDim.x - slab + nslab - 1) / nslab;
    const int n0 = slab * SLAB;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.B);
    T* __restrict__ C = static_cast<T*>(g.C);
    v8 wfr[NCT][KSH];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
            wfr[ct][ks] = *reinterpret_cast<const v8*>(W + (long)(n0 + p * (NCT * 16) + ct * 16 + l15) * g.ldb + h * 384 + ks * 32 + gq * 8);
    if (t < SLAB) s_bias[t] = g.bias ? g.bias[n0 + t] : 0.f;
    const long ntile = ((long)g.M + ROWS - 1) / ROWS;
    auto dma_tile = [&](long tile, int buf) {
        const long r0 = tile * ROWS;
#pragma unroll
        for (int m = 0; m < NDMA; ++m) {
            const unsigned q = (unsigned)((wave + 8 * m) * 64 + lane);
            const unsigned r = (q * 43691u) >> 22;                    // q / 96 for q < 4096
            const unsigned c = (q - r * 96u) ^ (r & 15u);
            long row = r0 + r;
            row = row < g.M ? row : (long)g.M - 1;
            wst_dma16(A + row * g.lda + c * 8u, &s_x[buf][(wave + 8 * m) * 512]);
        }
    };
    float rgmax = 0.f;
    long tile = stream;
    if (tile < ntile) dma_tile(tile, 0);
    int buf = 0;
    // the 16-bit results of a tile wait in six registers and are stored one tile LATER, behind the next DMA issue: everything the counted
    // wait at the top of an iteration covers (vmcnt(0): DMAs and stores alike) is then a whole tile old, nothing freshly issued is waited for
    t4 pend[NCT];
    long prow = -1;
    auto flush = [&]() {
        if (prow >= 0 && prow < g.M) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<t4*>(C + prow * g.ldc + n0 + p * (NCT * 16) + ct * 16 + gq * 4) = pend[ct];
        }
    };
    for (; tile < ntile; tile += nstream) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this tile's rows and the stores of the tile before the previous one
        WST_BAR();                                                     // (1) tile complete; everybody is done with the other buffer and with s_ex
        const long next = tile + nstream;
        if (next < ntile) dma_tile(next, buf ^ 1);
        flush();
        f4 acc[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* xb = &s_x[buf][0];
        auto xfrag = [&](int idx) {
            const int ks = idx / RT, rt = idx % RT;
            return *reinterpret_cast<const v8*>(xb + ((rt * 16 + l15) * CPR + ((h * 48 + ks * 4 + gq) ^ l15)) * 8);
        };
        v8 xq[4];
#pragma unroll
        for (int pre = 0; pre < 3; ++pre) xq[pre] = xfrag(pre);
#pragma unroll
        for (int idx = 0; idx < KSH * RT; ++idx) {
            if (idx + 3 < KSH * RT) xq[(idx + 3) & 3] = xfrag(idx + 3);
            const int ks = idx / RT, rt = idx % RT;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = mma16<T>(wfr[ct][ks], xq[idx & 3], acc[rt][ct]);
        }
        // the K halves meet: wave h finishes row tile h and hands row tile 1 - h to its partner
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f4*>(&s_ex[wave][ct][lane * 4]) = h ? acc[0][ct] : acc[1][ct];
        WST_BAR();                                                     // (2)
        const long row = tile * ROWS + h * 16 + l15;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f4 other = *reinterpret_cast<const f4*>(&s_ex[wave ^ 1][ct][lane * 4]);
            f4 v = (h ? acc[1][ct] : acc[0][ct]) + other;
            v = v + *reinterpret_cast<const f4*>(&s_bias[p * (NCT * 16) + ct * 16 + gq * 4]);
            if constexpr (GELU) v = gelu16_fast4(v);
            if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4_f(rgmax, v);
            pend[ct] = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        }
        prow = row;
        buf ^= 1;
    }
    flush();
    if constexpr (std::is_same<T, _Float16>::value) rg_report_f(rgmax, g.ovf, 3u);
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the product is one this schedule is built for: 16-bit output, K = 768, N a multiple of 256, at
// least as many row tiles as workgroups per slab.
int gemm16_wst(const G16Args& g, int out16, int precision, hipStream_t st) {
    if (!out16 || g.K != 768 || (g.N % 192) || g.resid || g.gamma || g.resid_period || g.rowtau || g.lnc_a || g.row_stats || g.ln16_out)
        return MI355_EUNSUPPORTED;
    if (g.act != MI355_ACT_NONE && g.act != MI355_ACT_GELU) return MI355_EUNSUPPORTED;
    if ((g.lda & 7) || (g.ldb & 7) || (g.ldc & 3) || !aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || (g.bias && !aligned16(g.bias)))
        return MI355_EUNSUPPORTED;
    if (precision != MI355_PREC_FP16 && precision != MI355_PREC_BF16) return MI355_EUNSUPPORTED;
    const int ncu = resident_slots(1);
    const int nslab = g.N / 192;
    if (nslab > ncu || (long)g.M < 32L * 8 * (ncu / nslab)) return MI355_EUNSUPPORTED;     // too few rows to amortise the resident weights
    MI355_TRACE(st, "gemm16_wst_kernel<%s,out16> M=%d N=%d K=%d%s", precision == MI355_PREC_FP16 ? "f16" : "bf16", g.M, g.N, g.K,
                g.act == MI355_ACT_GELU ? " gelu" : "");
    if (g.act == MI355_ACT_GELU) {
        if (precision == MI355_PREC_FP16) gemm16_wst_kernel<_Float16, true><<<ncu, 512, 0, st>>>(g, nslab);
        else                              gemm16_wst_kernel<__bf16, true><<<ncu, 512, 0, st>>>(g, nslab);
    } else {
        if (precision == MI355_PREC_FP16) gemm16_wst_kernel<_Float16, false><<<ncu, 512, 0, st>>>(g, nslab);
        else                              gemm16_wst_kernel<__bf16, false><<<ncu, 512, 0, st>>>(g, nslab);
    }
    return MI355_OK;
}

}  // namespace mi355
