// gemm16_w4.hip -- persistent 256 x 256 x 64 GEMM on the 16-bit engine with ONE WAVE PER SIMD: four waves 2 (M) x 2 (N), 128 x 128 outputs
// each, 256 accumulation registers (AGPRs) per lane, a hand-placed instruction stream.
//
//   Y16 (M x N) = act( X16 (M x K) . W16^T (N x K) + bias )       16-bit outputs only; same K order and bit-identical results as gemm16_p8.hip
//   (fp32 outputs with a residual stay on gemm16_pa.hip, whose second accumulator set hides their epilogue: measured)
//
// Why (tools/mfma_probe.hip, profiles/r05_mfma_probe.md): a 128 x 64 wave tile needs 12 fragment reads per 32 MFMAs (0.375 per MFMA) and two
// waves per SIMD that trade places at barriers; fed from LDS it sustains 1.38-1.46 PFLOP/s.  A 128 x 128 wave tile needs 16 reads per 64 MFMAs
// (0.25 per MFMA), and when ONE wave issues them between its own MFMAs -- the reads of the NEXT K = 32 step into a second register set, one
// LDS-DMA piece every fourth MFMA, one barrier per K-tile -- the matrix pipe is busy 91 % of the cycles (17.8 cycles per MFMA, 1.76 PFLOP/s at
// the clock the power limit leaves).  The compiler does not produce that stream (its own order serialises reads and MFMAs: 32 cycles per MFMA),
// so the main loop is a sequence of `asm volatile` statements whose ORDER is the schedule; registers are still allocated by the compiler.
//
// LDS: 2 K-tile buffers x 64 KB (A: 256 rows x 128 B, B: 256 rows x 128 B; 16-byte chunk c of row r sits at chunk c ^ (r & 7): the swizzle is
// applied on the SOURCE side of the DMA) + a 2 KB epilogue slab per wave.  Per K-tile T (buffer b = T & 1):
//   S0: 64 MFMAs on fragment set 0 (k = 0..31 of K-tile T); between them the 16 ds_read_b128 of set 1 (k = 32..63) from buffer b
//       s_waitcnt vmcnt(0)  [this wave's pieces of K-tile T + 1, issued half a K-tile ago or earlier]  s_barrier
//       -> every wave has read everything of buffer b (its set-1 reads were retired by the lgkmcnt(0) that closes S0), K-tile T + 1 is in LDS
//   S1: 64 MFMAs on set 1; between them the 16 reads of set 0 of K-tile T + 1 from buffer b ^ 1 and the wave's 16 DMA pieces
//       (1 KB each: rows wave * 64 + p * 8 .. + 7 of A, then of B) of K-tile T + 2 into buffer b
// The stream of K-tiles is continuous over the workgroup's output tiles; the first K = 32 step of a tile multiplies into a zero C operand
// (no accumulator clearing).  Epilogue: bias (staged into LDS by one DMA piece per tile) / GELU in the accumulator layout, transposed through
// two wave-private 2 KB slabs into whole 128-byte row lines; the slab reads of a step are consumed one step later (comment at the loop).
#include <type_traits>
#include "gemm16.h"
#include "bufops.h"

namespace {
using namespace g16;
typedef unsigned int w4_u4 __attribute__((ext_vector_type(4)));

struct W4Plan {
    int tiles_n;
    int full;      // whole rounds: every workgroup walks `full` tiles ...
    int left;      // ... and workgroups 0 .. left-1 one more
};

template <int I> struct W4I { static constexpr int v = I; };
template <int B, int N>
struct W4Unroll {
    template <class F>
    static __device__ __forceinline__ void run(F& f) {
        f(W4I<B>{});
        W4Unroll<B + 1, N>::run(f);
    }
};
template <int N>
struct W4Unroll<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F&) {}
};

__device__ __forceinline__ unsigned w4_lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}
// one 1 KB LDS-DMA piece: 16 B per lane from the wave-uniform base + one 32-bit lane offset, lane-linear at the wave-uniform LDS byte address
__device__ __forceinline__ void w4_dma(const void* base, unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(base), "s"(dst)
                 : "memory");
}
template <typename T, bool ZERO>
__device__ __forceinline__ void w4_mfma(f4& c, const w4_u4& b, const w4_u4& a) {
    if constexpr (std::is_same<T, _Float16>::value) {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(b), "v"(a));
        else                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
    } else {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(b), "v"(a));
        else                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
    }
}
// keeps an accumulator tile in its AGPRs up to this point of the epilogue (without it the register allocator copies all 256 accumulation
// registers into VGPRs right behind the main loop and spills)
__device__ __forceinline__ void w4_pin(f4& c) { asm volatile("" : "+a"(c)); }
template <int OFF>
__device__ __forceinline__ void w4_read(w4_u4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// retires every outstanding fragment read and NAMES the registers they fill: no use of them can be scheduled above this statement
__device__ __forceinline__ void w4_reads_done(w4_u4 (&fa)[8], w4_u4 (&fb)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]), "+v"(fb[0]), "+v"(fb[1]),
                   "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7])
                 :
                 : "memory");
}

// DMA state of one K-tile of the stream: wave-uniform byte pointers of this wave's first A / B row at the K-tile's first column
struct W4Src { const char* a; const char* b; };

// One K = 32 step: 64 MFMAs acc[i][j] += B-fragment j x A-fragment i of the CURRENT set (ca / cb), with the 16 reads of the NEXT set (na / nb,
// LDS byte addresses ra / rb + tile * 2048) after MFMAs 0, 2, ... 30 and -- DMA -- this wave's 16 pieces of a later K-tile after MFMAs 2, 6, ... 62.
template <typename T, bool ZERO, bool DMA>
__device__ __forceinline__ void w4_step(f4 (&acc)[64], const w4_u4 (&ca)[8], const w4_u4 (&cb)[8], w4_u4 (&na)[8], w4_u4 (&nb)[8], unsigned ra,
                                        unsigned rb, const W4Src& src, unsigned stride_a, unsigned stride_b, unsigned voff_a, unsigned voff_b,
                                        unsigned dst) {
    auto one = [&](auto ic) {
        constexpr int m = decltype(ic)::v, i = m >> 3, j = m & 7;
        w4_mfma<T, ZERO>(acc[m], cb[j], ca[i]);
        if constexpr ((m & 1) == 0 && m < 32) {
            constexpr int r = m >> 1;                                   // B fragments first: the next step opens with row tile 0 against all of them
            if constexpr (r < 8) w4_read<r * 2048>(nb[r], rb);
            else                 w4_read<(r - 8) * 2048>(na[r - 8], ra);
        }
        if constexpr (DMA && (m & 3) == 2) {
            constexpr int p = m >> 2;                                   // pieces 0-7: A rows p * 8 .., pieces 8-15: B rows (p - 8) * 8 ..
            if constexpr (p < 8) w4_dma(src.a + (size_t)p * stride_a, voff_a, dst + p * 1024);
            else                 w4_dma(src.b + (size_t)(p - 8) * stride_b, voff_b, dst + 32768 + (p - 8) * 1024);
        }
    };
    W4Unroll<0, 64>::run(one);
    w4_reads_done(na, nb);
}

template <typename T, bool GELU>
__global__ __launch_bounds__(256) void gemm16_w4_kernel(const G16Args g, const W4Plan pl) {
    using v4 = typename Vec8<T>::t4;
    constexpr unsigned BUFB = 65536u, SLAB0 = 131072u, BIAS0 = SLAB0 + 4u * 4096u;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[BIAS0 + 2048u];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = pl.tiles_n;
    const int nk = g.K / BK;
    const unsigned lds0 = w4_lds_addr(lds_raw);

    // ---- this workgroup's tile list (the dealing of gemm16_p8.hip: XCD x owns a contiguous range of the n-fastest tile order) ----------
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int gq = gridDim.x >> 3, gr = gridDim.x & 7;
    const int per_xcd = gq + (xcd < gr ? 1 : 0);
    const int my_first = pl.full * (xcd * gq + (xcd < gr ? xcd : gr)) + slot_in_xcd;
    const int my_count = pl.full + ((int)blockIdx.x < pl.left ? 1 : 0);
    if (my_count == 0) return;
    const int total_kt = my_count * nk;
    auto entry_tile = [&](int e) { return e < pl.full ? my_first + e * per_xcd : pl.full * (int)gridDim.x + (int)blockIdx.x; };

    // ---- DMA sources: one piece = 8 rows x 128 B; lane -> (row lane >> 3, physical chunk lane & 7) fetches logical chunk (lane & 7) ^ row ----
    const int lrow = lane >> 3;
    const unsigned csw = (unsigned)(((lane & 7) ^ lrow) * 16);
    const unsigned voff_a = (unsigned)(lrow * g.lda) * 2u + csw, voff_b = (unsigned)(lrow * g.ldb) * 2u + csw;
    const unsigned stride_a = (unsigned)g.lda * 16u, stride_b = (unsigned)g.ldb * 16u;         // 8 rows, bytes
    const char* Ab = static_cast<const char*>(g.A);
    const char* Bb = static_cast<const char*>(g.B);
    int c_e = 0, c_kt = 0;                                                // cursor of the next K-tile to stage
    W4Src cs;
    auto seek = [&](int e) {
        const int tile = entry_tile(e);
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        cs.a = Ab + ((size_t)(m0 + wave * 64) * g.lda) * 2;
        cs.b = Bb + ((size_t)(n0 + wave * 64) * g.ldb) * 2;
        c_e = e; c_kt = 0;
    };
    auto advance = [&]() {                                               // past the end the cursor stays on the last K-tile (staged again, never read)
        if (c_kt + 1 < nk) { ++c_kt; cs.a += 128; cs.b += 128; }
        else if (c_e + 1 < my_count) seek(c_e + 1);
    };
    auto stage_all = [&](unsigned dst) {                                 // prologue form: the 16 pieces back to back
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            w4_dma(cs.a + (size_t)p * stride_a, voff_a, dst + p * 1024);
            w4_dma(cs.b + (size_t)p * stride_b, voff_b, dst + 32768 + p * 1024);
        }
    };

    // ---- fragment addresses: row (tile * 16 + lane & 15) of the wave's 128 rows, chunk (kk * 4 + lane >> 4) ^ (lane & 7) ---------------
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    const unsigned offk[2] = {(unsigned)((fq ^ fsw) * 16), (unsigned)(((4 + fq) ^ fsw) * 16)};
    const unsigned ra_base = lds0 + (unsigned)((wr * 128 + frow) * 128), rb_base = lds0 + 32768u + (unsigned)((wc * 128 + frow) * 128);
    const unsigned dst_base = lds0 + (unsigned)(wave * 64 * 128);

    f4 acc[64];
    w4_u4 fa[2][8], fb[2][8];
    float rgmax = 0.f;

    if (!g.bias) reinterpret_cast<float*>(lds_raw + BIAS0)[t] = 0.f, reinterpret_cast<float*>(lds_raw + BIAS0)[256 + t] = 0.f;      // read as zeros by every epilogue

    // ---- prologue: K-tiles 0 and 1 into buffers 0 and 1, set 0 of K-tile 0 into registers ----------------------------------------------
    seek(0);
    stage_all(dst_base);
    advance();
    stage_all(dst_base + BUFB);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        auto rd = [&](auto ic) {
            constexpr int r = decltype(ic)::v;
            w4_read<r * 2048>(fb[0][r], rb_base + offk[0]);
            w4_read<r * 2048>(fa[0][r], ra_base + offk[0]);
        };
        W4Unroll<0, 8>::run(rd);
        w4_reads_done(fa[0], fb[0]);
    }

    int kt_in_tile = 0, out_tile = 0, tm0 = 0, tn0 = 0;
    for (int T_ = 0; T_ < total_kt; ++T_) {
        const unsigned b_cur = (unsigned)(T_ & 1) * BUFB, b_nxt = BUFB - b_cur;
        // S0: set 0 of K-tile T_; reads: set 1 of K-tile T_ (buffer b_cur)
        if (kt_in_tile == 0) {
            const int tile = entry_tile(out_tile);
            tm0 = (tile / tiles_n) * 256; tn0 = (tile % tiles_n) * 256;
            // bias of the tile's 256 columns: one DMA piece into the LDS area of this tile's parity (the epilogue reads it with ds_read: a
            // compiler-counted global load there would be answered with a vmcnt wait that also drains the DMA stream)
            if (wave == 0 && g.bias)
                w4_dma(reinterpret_cast<const char*>(g.bias) + (size_t)tn0 * 4, (unsigned)lane * 16u, lds0 + BIAS0 + (unsigned)(out_tile & 1) * 1024u);
            w4_step<T, true, false>(acc, fa[0], fb[0], fa[1], fb[1], ra_base + b_cur + offk[1], rb_base + b_cur + offk[1], cs, 0, 0, 0, 0, 0);
        } else {
            w4_step<T, false, false>(acc, fa[0], fb[0], fa[1], fb[1], ra_base + b_cur + offk[1], rb_base + b_cur + offk[1], cs, 0, 0, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // S1: set 1; reads: set 0 of K-tile T_ + 1 (buffer b_nxt); DMA: K-tile T_ + 2 into buffer b_cur
        w4_step<T, false, true>(acc, fa[1], fb[1], fa[0], fb[0], ra_base + b_nxt + offk[0], rb_base + b_nxt + offk[0], cs, stride_a, stride_b, voff_a,
                                     voff_b, dst_base + b_cur);
        advance();

        if (++kt_in_tile == nk) {
            // ---- epilogue of this output tile (transposed MFMA tiles: lane = one row, 4 consecutive columns) -----------------------------
            // A step = 16 rows x 128 bytes of output through one of the wave's two 2 KB slabs: accumulators -> bias / GELU / LayerScale /
            // convert -> slab (accumulator layout) -> two row-line reads.  The reads of step s are issued right behind its writes (a wave's LDS
            // operations execute in order) and CONSUMED during step s + 1, behind that step's VALU work: a lone wave per SIMD has nobody to hide
            // an LDS round trip behind (the unpipelined form of gemm16_p8.hip took 11.3 k cycles per tile here, 4.6 k of them instructions).
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");     // the last MFMAs' results: 18 wait states before a VALU reads them
            const int l15 = lane & 15, fq4 = lane >> 4;
            unsigned char* slab = lds_raw + SLAB0 + wave * 4096;
            const int srow = lane >> 3, sch = lane & 7;                  // read side: row srow (+8), 16-byte chunk sch
            const int mrow0 = tm0 + wr * 128, ncol0 = tn0 + wc * 128;
            const unsigned char* bia = lds_raw + BIAS0 + (out_tile & 1) * 1024 + (wc * 128 + fq4 * 4) * 4;
            constexpr unsigned ES = 2u;
            const unsigned span = (unsigned)(127 * g.ldc + 128) * ES;
            const rsrc_t rc = make_rsrc(static_cast<char*>(g.C) + ((size_t)mrow0 * g.ldc + ncol0) * ES, span);
            const unsigned row8 = (unsigned)(8 * g.ldc) * ES;          // bytes between the two row halves of a step
            {
                const unsigned vo = ((unsigned)(srow * g.ldc) + (unsigned)sch * 8u) * 2u;
                f4 bias4[4];
                w4_u4 o_prev[2] = {};
                unsigned so_prev = 0;
#pragma unroll
                for (int s = 0; s < 16; ++s) {                           // 64 columns (jh) x 16 rows (i) per step
                    const int jh = s >> 3, i = s & 7;
                    unsigned char* sl = slab + (s & 1) * 2048;
                    if (i == 0) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bias4[jj] = *reinterpret_cast<const f4*>(bia + (jh * 4 + jj) * 64);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        w4_pin(acc[i * 8 + jh * 4 + jj]);
                        f4 v = acc[i * 8 + jh * 4 + jj] + bias4[jj];
                        if constexpr (GELU) {
                            v = gelu16_fast4(v);
                            asm("" : "+v"(v));                          // the fp32 value is ROUNDED before it is converted, as in every other kernel of the engine
                        }                                               // (left alone, hipcc fuses the last FMA of the GELU with the conversion: v_fma_mixlo_f16, one rounding)
                        if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4(rgmax, v);
                        *reinterpret_cast<v4*>(sl + l15 * 128 + (((jj * 2 + (fq4 >> 1)) ^ (l15 & 7)) * 16) + (fq4 & 1) * 8) = v4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    w4_u4 o_cur[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = h * 8 + srow;
                        o_cur[h] = *reinterpret_cast<const w4_u4*>(sl + r * 128 + ((sch ^ (r & 7)) * 16));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (s > 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            __builtin_amdgcn_raw_buffer_store_b128(o_prev[h], rc, vo, so_prev + (unsigned)h * row8, 0);
                        }
                    }
                    o_prev[0] = o_cur[0]; o_prev[1] = o_cur[1];
                    so_prev = ((unsigned)(i * 16 * g.ldc) + (unsigned)jh * 64u) * 2u;
                    __builtin_amdgcn_sched_barrier(0);                  // one step at a time: the accumulators leave the AGPRs as they are needed
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) __builtin_amdgcn_raw_buffer_store_b128(o_prev[h], rc, vo, so_prev + (unsigned)h * row8, 0);
            }
            kt_in_tile = 0;
            ++out_tile;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the re-staged pieces past the end of the stream
    if constexpr (std::is_same<T, _Float16>::value) rg_report(rgmax, g.ovf, 3u);
}

}  // namespace

namespace mi355 {

// Launch the one-wave-per-SIMD persistent kernel when the shape suits it (whole 256 x 256 tiles, at least one round of them, K >= 128).
// Returns MI355_EUNSUPPORTED without touching anything when it does not apply.
int gemm16_w4(const g16::G16Args& g, int out16, int precision, hipStream_t st) {
    if ((g.M & 255) || (g.N & 255) || (g.K % g16::BK) || g.K < 2 * g16::BK) return MI355_EUNSUPPORTED;
    if (!out16 || g.resid || g.gamma || g.resid_period || g.rowtau || g.lnc_a) return MI355_EUNSUPPORTED;      // fp32 (+ residual) outputs: gemm16_pa hides their epilogue
    if ((g.lda & 7) || (g.ldb & 7) || !aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || (g.ldc & 7)) return MI355_EUNSUPPORTED;
    if ((g.bias && !aligned16(g.bias)) || (g.gamma && !aligned16(g.gamma)) || (g.resid && !aligned16(g.resid))) return MI355_EUNSUPPORTED;
    if ((size_t)g.lda * 16u >= (1ull << 31) || (size_t)g.ldb * 16u >= (1ull << 31)) return MI355_EUNSUPPORTED;
    const long ntiles = (long)(g.M / 256) * (g.N / 256);
    if (ntiles > (1L << 30)) return MI355_EUNSUPPORTED;
    const int ncu = resident_slots(1);
    W4Plan pl{};
    pl.tiles_n = g.N / 256;
    const int grid = ntiles < ncu ? (int)ntiles : ncu;
    pl.full = (int)(ntiles / grid);
    pl.left = (int)(ntiles - (long)pl.full * grid);
    MI355_TRACE(st, "gemm16_w4_kernel<%s,out16> M=%d N=%d K=%d%s", precision == MI355_PREC_FP16 ? "f16" : "bf16", g.M, g.N, g.K,
                g.act == MI355_ACT_GELU ? " gelu" : "");
    if (g.act == MI355_ACT_GELU) {
        if (precision == MI355_PREC_FP16) gemm16_w4_kernel<_Float16, true><<<grid, 256, 0, st>>>(g, pl);
        else                              gemm16_w4_kernel<__bf16, true><<<grid, 256, 0, st>>>(g, pl);
    } else {
        if (precision == MI355_PREC_FP16) gemm16_w4_kernel<_Float16, false><<<grid, 256, 0, st>>>(g, pl);
        else                              gemm16_w4_kernel<__bf16, false><<<grid, 256, 0, st>>>(g, pl);
    }
    return MI355_OK;
}

}  // namespace mi355
