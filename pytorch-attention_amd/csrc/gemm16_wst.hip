// gemm16_wst.hip -- Y16 (M x N, 16 bit) = act(X16 (M x 768) . W16^T (N x 768) + bias) with the WEIGHTS STATIONARY in registers: the qkv and fc1
// products of a ViT-Base encoder layer (ViT.py:81 and :59-61; N = 2304 / 3072, K = 768, M = 50 432 at B = 256), gfx950.  Round 6.
//
// The tile kernels of the engine (gemm16_p8 / gemm16_w4 / gemm16_pa) stream BOTH operands through LDS for every 256 x 256 output tile:
// 1.39 GB cross the L2 -> CU path for the qkv product, half of it the same 3.5 MB of weights over and over, and the MFMA pipes sit 0.44-0.53
// busy (profiles/r06_c5_mfma_util.txt).  With K = 768 a 256-column slab of W is 393 KB -- it FITS the register file of a CU:
//
//   workgroup  = 8 waves, persistent, owns ONE 256-column slab for the whole kernel and walks 16-row tiles of X (the workgroups of a slab
//                take the row tiles round-robin; the slabs of one row tile sit on one XCD, so X crosses HBM -> L2 once per XCD);
//   wave (p, h) p = wave >> 1: columns [64 p, 64 p + 64) of the slab; h = wave & 1: the K half [384 h, 384 h + 384).  Its 64 x 384 block of W
//                lives in VGPRs as MFMA A-fragments (4 column tiles x 12 k-steps x 4 = 192 registers), loaded once;
//   X tile       16 rows x 768 = 24 KB, LDS-DMA'ed as one linear block (16-byte chunks XOR-swizzled at the SOURCE: chunk c of row r sits at
//                chunk c ^ r, so the 16 rows a fragment read touches fall into 16 different bank groups), double-buffered: the next tile is in
//                flight during the MFMAs; an X fragment read from LDS feeds four MFMAs;
//   per tile     48 MFMAs per wave; the two K halves of a column block are added through LDS (each wave hands over the two column tiles it
//                does not own and finishes the other two: bias, GELU, 16 bit, 64-byte row segments); two raw barriers.
// L2 -> CU bytes: X once per slab (9 x 77 MB for qkv) + W once per workgroup, against 1.39 GB for the 256 x 256 tiling.
// A row's K steps are added as (k < 384) + (k >= 384), each half in ascending order: NOT the bit pattern of the tile kernels (one chain), but
// the same for a row wherever it sits in the batch.  Option "gemm_wst".
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

template <typename T, bool GELU>
__global__ __launch_bounds__(512, 1) void gemm16_wst_kernel(const G16Args g, int nslab) {
    using v8 = typename Vec8<T>::t;
    typedef T t4 __attribute__((ext_vector_type(4)));
    constexpr int K = 768, KSH = 12, ROWS = 16, CPR = K / 8;           // 96 16-byte chunks per row
    constexpr int TILE_EL = ROWS * K;                                 // elements of an X tile (24 KB)
    constexpr int NDMA = TILE_EL * 2 / 1024 / 8;                      // 1 KB DMA instructions per wave and tile: 3
    __shared__ __attribute__((aligned(1024))) unsigned short s_x[2][TILE_EL];
    __shared__ __attribute__((aligned(16))) float s_ex[8][2][64 * 4];  // per wave: the two column tiles it hands to its partner
    __shared__ __attribute__((aligned(16))) float s_bias[256];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, gq = lane >> 4;
    const int p = wave >> 1, h = wave & 1;
    const int lid = xcd_contiguous_block();
    const int slab = lid % nslab, stream = lid / nslab;
    const int nstream = ((int)gridDim.x - slab + nslab - 1) / nslab;   // workgroups that share this slab
    const int n0 = slab * 256;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.B);
    T* __restrict__ C = static_cast<T*>(g.C);

    // ---- this wave's 64 x 384 block of W: A-operand fragments (row = output column, k = 384 h + 32 ks + 8 gq + [0,8)) ---------------------
    v8 wfr[4][KSH];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
            wfr[ct][ks] = *reinterpret_cast<const v8*>(W + (long)(n0 + p * 64 + ct * 16 + l15) * g.ldb + h * 384 + ks * 32 + gq * 8);
    if (t < 256) s_bias[t] = g.bias ? g.bias[n0 + t] : 0.f;

    // ---- DMA geometry: instruction j = wave + 8 m moves LDS chunks 64 j .. 64 j + 63; LDS chunk q = (row q / 96, slot q % 96) holds the row's
    //      chunk slot ^ row.  Recomputed per tile (a multiply-high and two integer ops per instruction) instead of held in six registers -----------
    const long ntile = ((long)g.M + ROWS - 1) / ROWS;
    auto dma_tile = [&](long tile, int buf) {
        const long r0 = tile * ROWS;
#pragma unroll
        for (int m = 0; m < NDMA; ++m) {
            const unsigned q = (unsigned)((wave + 8 * m) * 64 + lane);
            const unsigned r = (q * 683u) >> 16;                      // q / 96 for q < 1536
            const unsigned c = (q - r * 96u) ^ r;                      // slot ^ row stays inside its group of 16 chunks
            long row = r0 + r;
            row = row < g.M ? row : (long)g.M - 1;                     // rows beyond M: any valid row (their outputs are never stored)
            wst_dma16(A + row * g.lda + c * 8u, &s_x[buf][(wave + 8 * m) * 512]);
        }
    };
    // fragment (ks) of the tile's row l15 for this K half: chunk 48 h + 4 ks + gq of the row, at slot chunk ^ l15
    const int xrow = l15 * CPR;
    float rgmax = 0.f;

    long tile = stream;
    if (tile < ntile) dma_tile(tile, 0);
    int buf = 0;
    for (; tile < ntile; tile += nstream) {
        const long next = tile + nstream;
        if (next < ntile) {
            dma_tile(next, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");           // all but the three just issued: this tile's rows have landed (loads return in order)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        WST_BAR();                                                     // (1) everybody's part of the tile is in LDS; s_ex is free again
        // ---- 48 MFMAs: Y^T tiles = W . X^T; lane (l15, gq) holds columns ct * 16 + gq * 4 + [0,4) of row l15 -------------------------------------
        f4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* xb = &s_x[buf][0];
        auto xfrag = [&](int ks) { return *reinterpret_cast<const v8*>(xb + (xrow + ((h * 48 + ks * 4 + gq) ^ l15)) * 8); };
        v8 x0 = xfrag(0), x1 = x0;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            if (ks + 1 < KSH) x1 = xfrag(ks + 1);                      // one k-step (four MFMAs of this wave, four of its SIMD partner) ahead
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = mma16<T>(wfr[ct][ks], x0, acc[ct]);
            x0 = x1;
        }
        // ---- the two K halves meet: this wave finishes column tiles 2 h, 2 h + 1 and hands the other two to its partner -----------------------
        const f4 own[2] = {h ? acc[2] : acc[0], h ? acc[3] : acc[1]};
        *reinterpret_cast<f4*>(&s_ex[wave][0][lane * 4]) = h ? acc[0] : acc[2];
        *reinterpret_cast<f4*>(&s_ex[wave][1][lane * 4]) = h ? acc[1] : acc[3];
        WST_BAR();                                                     // (2) partials published; everybody is done reading s_x[buf]
        const long row = tile * ROWS + l15;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ct = 2 * h + k;
            const f4 other = *reinterpret_cast<const f4*>(&s_ex[wave ^ 1][k][lane * 4]);
            f4 v = own[k] + other;                                     // (k < 384) + (k >= 384): fp32 addition is commutative, both owners form the same sum
            v = v + *reinterpret_cast<const f4*>(&s_bias[p * 64 + ct * 16 + gq * 4]);
            if constexpr (GELU) v = gelu16_fast4(v);
            if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4_f(rgmax, v);
            if (row < g.M)
                *reinterpret_cast<t4*>(C + row * g.ldc + n0 + p * 64 + ct * 16 + gq * 4) = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        }
        buf ^= 1;
    }
    if constexpr (std::is_same<T, _Float16>::value) rg_report_f(rgmax, g.ovf, 3u);
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the product is one this schedule is built for: 16-bit output, K = 768, N a multiple of 256, at
// least as many row tiles as workgroups per slab.
int gemm16_wst(const G16Args& g, int out16, int precision, hipStream_t st) {
    if (!out16 || g.K != 768 || (g.N & 255) || g.resid || g.gamma || g.resid_period || g.rowtau || g.lnc_a || g.row_stats || g.ln16_out)
        return MI355_EUNSUPPORTED;
    if (g.act != MI355_ACT_NONE && g.act != MI355_ACT_GELU) return MI355_EUNSUPPORTED;
    if ((g.lda & 7) || (g.ldb & 7) || (g.ldc & 3) || !aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || (g.bias && !aligned16(g.bias)))
        return MI355_EUNSUPPORTED;
    if (precision != MI355_PREC_FP16 && precision != MI355_PREC_BF16) return MI355_EUNSUPPORTED;
    const int ncu = resident_slots(1);
    const int nslab = g.N / 256;
    if (nslab > ncu || (long)g.M < 16L * 8 * (ncu / nslab)) return MI355_EUNSUPPORTED;     // too few rows to amortise the resident weights
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
