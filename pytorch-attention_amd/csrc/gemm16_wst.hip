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
// row counts the one-wave-per-SIMD kernel below does not take (M % 32 != 0: 192-200 vs 192-204 us).  Two barriers, the K-half exchange and the
// per-tile DMA address arithmetic leave the matrix pipes 0.43 busy -- below the tile kernels' 0.53.  Option "gemm_wst" = 1 / 2 (default 0).
// gemm16_wst1_kernel further down ("gemm_wst" = 3 / 4) is the one-wave-per-SIMD form with W in AGPRs: bit-identical to the tile kernels, qkv
// 200-217 us, fc1 318-337 (281 with undeferred stores): a tie at best, also opt-in.
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
// (Measured and removed: 256-column slabs with 16-row tiles -- 192 registers of W, twice the barriers per row: 244-262 us for the qkv product.
// profiles/r06_gemm_wst.md)
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

// ---- ONE wave per SIMD: wave p owns 64 columns of a 256-column slab for the WHOLE reduction -- 64 x 768 of W = 384 registers, the first sixteen
// k-steps in AGPRs (256) and the last eight in VGPRs (128).  hipcc will not keep MFMA A-operands in AGPRs on its own (it shuttles them through
// v_accvgpr moves: 545 us, profiles/r06_gemm_wst.md), so the MFMAs are inline assembly whose A-operand constraint is "a": the fragments are
// written to their accumulation registers once and read there by the matrix pipe for the rest of the kernel.  No K split (a row's K steps form one
// ascending chain: the bit pattern of the tile kernels), no exchange, ONE barrier per 32-row tile, a ring of three X tiles, stores deferred by one
// tile.  A tile takes 3.6-3.8 us of which the MFMAs are 1.6: one wave per SIMD has nobody to hide its serial work behind (profiles/r06_gemm_wst.md).
template <typename T>
__device__ __forceinline__ void wst_mfma_a(f4& c, const typename Vec8<T>::t& w, const typename Vec8<T>::t& x) {
    if constexpr (std::is_same<T, _Float16>::value) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
    else                                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
}
template <typename T>
__device__ __forceinline__ void wst_mfma_v(f4& c, const typename Vec8<T>::t& w, const typename Vec8<T>::t& x) {
    if constexpr (std::is_same<T, _Float16>::value) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
    else                                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
// one 1 KB LDS-DMA piece from a wave-uniform base + a 32-bit lane offset (no 64-bit address arithmetic per piece)
__device__ __forceinline__ void wst_dma_so(const void* base, unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(dst)
                 : "memory");
}

template <typename T, bool GELU>
__global__ __launch_bounds__(256, 1) void gemm16_wst1_kernel(const G16Args g, int nslab) {
    using v8 = typename Vec8<T>::t;
    typedef T t4 __attribute__((ext_vector_type(4)));
    constexpr int K = 768, KS = 24, KSA = 16, RT = 2, ROWS = 16 * RT, CPR = K / 8;
    constexpr int TILE_EL = ROWS * K;                                 // 48 KB
    constexpr int NDMA = TILE_EL * 2 / 1024 / 4;                      // 12 pieces per wave and tile
    constexpr int NB = 3;
    __shared__ __attribute__((aligned(1024))) unsigned short s_x[NB][TILE_EL];
    __shared__ __attribute__((aligned(16))) float s_bias[256];
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lid = xcd_contiguous_block();
    const int slab = lid % nslab, stream = lid / nslab;
    const int nstream = ((int)gridDim.x - slab + nslab - 1) / nslab;
    const int n0 = slab * 256 + wave * 64;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.B);
    T* __restrict__ C = static_cast<T*>(g.C);
    // ---- W fragments: k-steps 0 .. 15 -> AGPRs (through the "a" constraint of their consumers), 16 .. 23 -> VGPRs -----------------------------
    v8 wa[4][KSA], wv[4][KS - KSA];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const T* wr = W + (long)(n0 + ct * 16 + l15) * g.ldb + gq * 8;
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) wa[ct][ks] = *reinterpret_cast<const v8*>(wr + ks * 32);
#pragma unroll
        for (int ks = KSA; ks < KS; ++ks) wv[ct][ks - KSA] = *reinterpret_cast<const v8*>(wr + ks * 32);
    }
    s_bias[t] = g.bias ? g.bias[slab * 256 + t] : 0.f;
    // ---- DMA: piece j = wave + 4 m moves LDS chunks 64 j .. 64 j + 63; LDS chunk q = (row q / 96, slot q % 96) holds the row's chunk slot ^ (row & 15) ----
    unsigned doff[NDMA];
#pragma unroll
    for (int m = 0; m < NDMA; ++m) {
        const unsigned q = (unsigned)((wave + 4 * m) * 64 + lane);
        const unsigned r = (q * 43691u) >> 22;                        // q / 96 for q < 4096
        const unsigned c = (q - r * 96u) ^ (r & 15u);
        doff[m] = (r * (unsigned)g.lda + c * 8u) * 2u;
    }
    const unsigned lds0 = wst_lds_addr(&s_x[0][0]);
    const long ntile = (long)g.M / ROWS;                              // launcher: M % 32 == 0
    auto dma_tile = [&](long tile, int buf) {
        const char* base = reinterpret_cast<const char*>(A + tile * ROWS * (long)g.lda);
#pragma unroll
        for (int m = 0; m < NDMA; ++m) wst_dma_so(base, doff[m], lds0 + (unsigned)buf * (TILE_EL * 2) + (unsigned)(wave + 4 * m) * 1024u);
    };
    float rgmax = 0.f;
    long tile = stream;
#pragma unroll
    for (int pre = 0; pre < NB - 1; ++pre)
        if (tile + (long)pre * nstream < ntile) dma_tile(tile + (long)pre * nstream, pre);
    int buf = 0;
    // The 16-bit results of a tile wait in registers and are stored one tile LATER, right behind the next DMA issue.  The counted wait at the
    // top of an iteration then sees, besides the tile it needs (two tiles old), only operations that are one whole tile old -- twelve DMA pieces
    // and eight stores: vmcnt(12) is met without waiting for anything fresh, and it is SAFE whatever order stores and loads complete in
    // (loads return in order: with at most twelve operations out, the pieces of the tile before the newest have all landed).  With the stores
    // issued at the end of the iteration the same wait stalled on them every tile (213-229 us for the qkv product).
    t4 pend[RT][4];
    long prow = -1;
    auto flush = [&]() {
        if (prow >= 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<t4*>(C + (prow + rt * 16) * g.ldc + n0 + ct * 16 + gq * 4) = pend[rt][ct];
        }
    };
    for (; tile < ntile; tile += nstream) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        WST_BAR();                                                     // the tile is complete in LDS; everybody is done with the previous one
        const long ahead = tile + (long)(NB - 1) * nstream;
        if (ahead < ntile) dma_tile(ahead, (buf + NB - 1) % NB);      // into the buffer the previous tile has just left
        flush();
        f4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* xb = &s_x[buf][0];
        auto xfrag = [&](int idx) {                                   // flat order over (ks, rt)
            const int ks = idx / RT, rt = idx % RT;
            return *reinterpret_cast<const v8*>(xb + ((rt * 16 + l15) * CPR + ((ks * 4 + gq) ^ l15)) * 8);
        };
        v8 xq[5];
#pragma unroll
        for (int pre = 0; pre < 4; ++pre) xq[pre] = xfrag(pre);
#pragma unroll
        for (int idx = 0; idx < KS * RT; ++idx) {
            if (idx + 4 < KS * RT) xq[(idx + 4) % 5] = xfrag(idx + 4);
            const int ks = idx / RT, rt = idx % RT;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ks < KSA) wst_mfma_a<T>(acc[rt][ct], wa[ct][ks < KSA ? ks : 0], xq[idx % 5]);
                else          wst_mfma_v<T>(acc[rt][ct], wv[ct][ks >= KSA ? ks - KSA : 0], xq[idx % 5]);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");             // the last MFMAs retire before the VALU reads their accumulators
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f4 v = acc[rt][ct] + *reinterpret_cast<const f4*>(&s_bias[wave * 64 + ct * 16 + gq * 4]);
                if constexpr (GELU) v = gelu16_fast4(v);
                if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4_f(rgmax, v);
                pend[rt][ct] = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
            }
        }
        prow = tile * ROWS + l15;
        buf = (buf + 1) % NB;
    }
    flush();
    if constexpr (std::is_same<T, _Float16>::value) rg_report_f(rgmax, g.ovf, 3u);
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the product is one this schedule is built for: 16-bit output, K = 768, N a multiple of 192 (options
// 3 / 4: of 256, and M of 32), at least eight row tiles per workgroup of a slab.
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
    if (opt_gemm_wst() >= 3) {                                         // one wave per SIMD, weight fragments in AGPRs (256-column slabs, M % 32 == 0)
        if ((g.N & 255) || (g.M & 31) || (long)g.lda * 32 * 2 >= (1L << 31)) return MI355_EUNSUPPORTED;
        const int ns = g.N / 256;
        if (g.act == MI355_ACT_GELU) {
            if (precision == MI355_PREC_FP16) gemm16_wst1_kernel<_Float16, true><<<ncu, 256, 0, st>>>(g, ns);
            else                              gemm16_wst1_kernel<__bf16, true><<<ncu, 256, 0, st>>>(g, ns);
        } else {
            if (precision == MI355_PREC_FP16) gemm16_wst1_kernel<_Float16, false><<<ncu, 256, 0, st>>>(g, ns);
            else                              gemm16_wst1_kernel<__bf16, false><<<ncu, 256, 0, st>>>(g, ns);
        }
        return MI355_OK;
    }
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
