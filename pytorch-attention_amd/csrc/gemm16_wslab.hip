// gemm16_wslab.hip -- Y16 (M x N, 16 bit) = act(X16 (M x K) . W16^T (N x K) + bias) for SHORT reductions (K = 256 / 384 / 512) with a column slab
// of W stationary in registers: the qkv and fc1 products of the token widths 256 ... 512 (XCiT xcit.py:251,:292; CSWin stages 3 / 4
// cswin.py:187,:195; the Mixer's channel MLP mlp_mixer.py:49), gfx950.  Round 6.
//
// gemm16_wreg.hip's dataflow (N = K, fp32 + residual output) widened to N = several slabs and given the 16-bit epilogue.  Why: with 4 ... 8 K-tiles the
// tile kernels (gemm16_pa) spend as long in a tile's epilogue as in its main loop and round M x N up to 128 x 256 tiles (CSWin stage 4: 588
// tiles on 256 workgroups = 0.77 full); they reach 0.43-0.74 PFLOP/s on these shapes (profiles/r06_*_kernel_seq.txt) against 1.0 at K = 768.
//
//   workgroup  = NW waves (4 at K = 256 / 384: two workgroups per CU with their own barriers; 8 at K = 512: one), persistent,
//                owns ONE slab of NW x NT x 16 columns for the whole kernel: wave w holds the NT x 16 columns [w NT 16, (w + 1) NT 16) of the slab
//                x the whole K as MFMA A-operand fragments (NT x K/32 x 4 registers: 128 ... 192), loaded once;
//   stream     = the workgroups of the slabs 0 .. nslab-1 with the same stream index sit next to each other on one XCD and walk the SAME 32-row
//                tiles of X (stream, stream + nstream, ...) at the same pace: X crosses HBM -> L2 once, L2 -> LDS once per slab;
//   row tile   = 32 rows of X in LDS (double-buffered through registers: the next tile's rows are in flight during the MFMAs), read by every wave
//                as B-operand fragments; 2 x NT x K/32 MFMAs per wave and tile, ONE barrier per tile;
//   columns    : MFMA row i = 4 q + r of column tile nt is slab column q (4 NT) + 4 nt + r -- a permutation of which W row sits in which
//                fragment slot, so that a lane's NT accumulator quads are 4 NT CONSECUTIVE output columns: one 8 NT-byte piece per row tile and
//                lane, the four lane groups of a wave complete 32 NT-byte row segments (128 B at NT = 4).
// A row's K steps are added in ascending order on the same MFMA instruction and the epilogue is gelu16_fast(acc + bias): the bits of the tile
// kernels (tests/test_round6_kernels_gpu.py).  Rows beyond M come back as zeros from the buffer range check and are never stored.
// MEASURED (profiles/r06_gemm_wslab.md, same process, interleaved): GELU epilogues 5-12 % faster than gemm16_pa (XCiT fc1 97-119 -> 89-105 us, CSWin
// stage-3 fc1 57 -> 51, stage-4 fc1 44 -> 40, Mixer fc1 131-150 -> 128-136), plain epilogues a tie (XCiT qkv 57-67 vs 59-71, CSWin s3 qkv 31 vs 32) or
// slower (CSWin s4 qkv 28.5 vs 31), row counts off the 256-row grid 88-96 -> 65-69 (those fall from gemm16_pa to gemm16_p8).  Hence option
// "gemm_wslab" = 1 (default): GELU epilogues and M % 256 != 0; 2: every product it takes; 0: none.  A ring of three LDS-DMA'ed tiles instead of the
// register staging was built and measured 3-8 % SLOWER on the GELU shapes (removed); the ablation says the kernel is bound by its epilogue VALU
// work + LDS fragment reads, not by the X stream.
#include "gemm16.h"
#include "bufops.h"
#include <type_traits>

namespace {

using namespace g16;

template <typename T, int K, int NT, int NW, bool GELU, bool A32 = false>     // A32: X arrives as fp32 rows (g.Af) and is rounded to T on its way into LDS
__global__ __launch_bounds__(NW * 64, 8 / NW) void gemm16_wslab_kernel(const G16Args g, int nslab, int nstream) {
    using v8 = typename Vec8<T>::t;
    typedef T t4 __attribute__((ext_vector_type(4)));
    constexpr int NTHR = NW * 64, RT = 2, ROWS = RT * 16;
    constexpr int KS = K / 32;
    constexpr int NCW = NT * 16;                  // columns per wave
    constexpr int SLAB = NW * NCW;
    constexpr int XP = K + 8;                     // LDS row pitch (elements)
    __shared__ __attribute__((aligned(16))) unsigned short s_x[2][ROWS * XP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, gq = lane >> 4;
    const int lid = xcd_contiguous_block();
    const int slab = lid % nslab, stream = lid / nslab;
    if (stream >= nstream) return;                // the workgroups beyond the last whole stream stay idle (whole workgroup: no barrier is left waiting)
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.B);
    T* __restrict__ Y = static_cast<T*>(g.C);
    const int n0 = slab * SLAB + wave * NCW;

    // ---- this wave's slice of W: A-operand fragments; fragment row l15 of column tile nt = column n0 + (l15 >> 2) (4 NT) + 4 nt + (l15 & 3) -------
    v8 wfr[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const T* wr = W + (long)(n0 + (l15 >> 2) * (4 * NT) + 4 * nt + (l15 & 3)) * g.ldb + gq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wfr[nt][ks] = *reinterpret_cast<const v8*>(wr + ks * 32);
    }
    const int ncol = n0 + gq * (4 * NT);          // this lane's 4 NT consecutive output columns
    f4 bias4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias4[nt] = g.bias ? *reinterpret_cast<const f4*>(g.bias + ncol + 4 * nt) : f4{0.f, 0.f, 0.f, 0.f};

    float rgmax = 0.f;
    const long ntile = ((long)g.M + ROWS - 1) / ROWS;
    // chunk c of a row tile: 16 bytes of row c / CPR16 -- eight 16-bit elements, or (A32) four fp32 elements that become 8 bytes of the LDS image
    constexpr int EPC = A32 ? 4 : 8, NCH = ROWS * (K / EPC) / NTHR;
    static_assert((ROWS * (K / EPC)) % NTHR == 0, "row tile must split evenly over the workgroup");
    int crow[NCH], ccol[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = t + j * NTHR;
        crow[j] = c / (K / EPC);
        ccol[j] = (c % (K / EPC)) * EPC;
    }
    typename std::conditional<A32, f4, v8>::type areg[NCH];
    float rgin = 0.f;                             // A32: the range guard of the conversion (what mi355_cast16_fwd reports)
    auto fetch = [&](long tile) {
        const long r0 = tile * ROWS;
        const long left = (long)g.M - r0;
        const int rows = (int)(left < ROWS ? left : ROWS);
        if constexpr (A32) {
            const rsrc_t rs = make_rsrc(g.Af + r0 * g.lda, (bufops_u32)(((long)(rows - 1) * g.lda + K) * 4));
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                areg[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (bufops_u32)((crow[j] * g.lda + ccol[j]) * 4), 0, 0));
        } else {
            const rsrc_t rs = make_rsrc(A + r0 * g.lda, (bufops_u32)(((long)(rows - 1) * g.lda + K) * 2));
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                areg[j] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rs, (bufops_u32)((crow[j] * g.lda + ccol[j]) * 2), 0, 0));
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if constexpr (A32) {
                const f4 v = areg[j];
                if constexpr (std::is_same<T, _Float16>::value) rgin = rg_absmax4(rgin, v);
                *reinterpret_cast<t4*>(&s_x[buf][crow[j] * XP + ccol[j]]) = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
            } else {
                *reinterpret_cast<v8*>(&s_x[buf][crow[j] * XP + ccol[j]]) = areg[j];
            }
        }
    };

    long tile = stream;
    if (tile < ntile) {
        fetch(tile);
        commit(0);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < ntile; tile += nstream) {
        const long r0 = tile * ROWS;
        const long left = (long)g.M - r0;
        const int rows = (int)(left < ROWS ? left : ROWS);
        const long next = tile + nstream;
        if (next < ntile) fetch(next);
        // ---- Y^T tiles = W . X^T: lane (l15, gq) holds columns ncol + 4 nt + [0,4) of row rt*16 + l15 --------------------------------------------
        f4 acc[RT][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f4{0.f, 0.f, 0.f, 0.f};
        // row tile by row tile: the epilogue arithmetic of row tile 0 has no dependence on the MFMAs of row tile 1 and is scheduled among them
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v8 xf = *reinterpret_cast<const v8*>(&s_x[buf][(rt * 16 + l15) * XP + ks * 32 + gq * 8]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = mma16<T>(wfr[nt][ks], xf, acc[rt][nt]);
            }
        }
        if (next < ntile) commit(buf ^ 1);
        // ---- epilogue: act(acc + bias) -> 16 bit, one 8 NT-byte piece per row tile and lane ----------------------------------------------------------
        const rsrc_t ry = make_rsrc(Y + r0 * g.ldc, (bufops_u32)(((long)(rows - 1) * g.ldc + g.N) * 2));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            t4 h[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f4 v = acc[rt][nt] + bias4[nt];
                if constexpr (GELU) v = gelu16_fast4(v);
                if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4_f(rgmax, v);
                h[nt] = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
            }
            const bufops_u32 off = (bufops_u32)(((rt * 16 + l15) * g.ldc + ncol) * 2);
            typedef unsigned int u2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int p = 0; p + 1 < NT; p += 2) {
                const u2 a = __builtin_bit_cast(u2, h[p]), b = __builtin_bit_cast(u2, h[p + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(u4{a.x, a.y, b.x, b.y}, ry, off + p * 8, 0, 0);
            }
            if constexpr (NT & 1) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, h[NT - 1]), ry, off + (NT - 1) * 8, 0, 0);
        }
        __syncthreads();                              // next tile's rows complete in s_x[buf ^ 1]; everybody is done reading s_x[buf]
        buf ^= 1;
    }
    if constexpr (std::is_same<T, _Float16>::value) {
        rg_report_f(rgmax, g.ovf, 3u);
        if constexpr (A32) rg_report(rgin, g.ovf, 1u);
    }
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the product is one this schedule is built for: 16-bit output, bias / GELU epilogue only,
// K = 256 / 384 / 512, N a multiple of the slab width of that K, enough rows that every stream walks several tiles.
namespace {
struct WslabPlan { int nw, nt, nslab, nstream, slots; bool a32; };
// MI355_OK + the launch geometry, or MI355_EUNSUPPORTED; touches nothing
int wslab_plan(const G16Args& g, int out16, int precision, WslabPlan* p) {
    if (!out16 || g.resid || g.gamma || g.resid_period || g.rowtau || g.lnc_a || g.row_stats || g.ln16_out) return MI355_EUNSUPPORTED;
    if (g.act != MI355_ACT_NONE && g.act != MI355_ACT_GELU) return MI355_EUNSUPPORTED;
    if (precision != MI355_PREC_FP16 && precision != MI355_PREC_BF16) return MI355_EUNSUPPORTED;
    const bool a32 = g.Af != nullptr;                // X as fp32 rows: the cast rides in the staging (mi355_linear16_x32_fwd)
    if ((g.lda & (a32 ? 3 : 7)) || (g.ldb & 7) || (g.ldc & 7) || g.ldb < g.K || g.lda < g.K || g.ldc < g.N) return MI355_EUNSUPPORTED;
    if (!aligned16(a32 ? static_cast<const void*>(g.Af) : g.A) || !aligned16(g.B) || !aligned16(g.C) || (g.bias && !aligned16(g.bias)))
        return MI355_EUNSUPPORTED;
    if ((long)32 * g.ldc * 2 >= (1L << 31) || (long)32 * g.lda * 4 >= (1L << 31)) return MI355_EUNSUPPORTED;
    // measured: K = 512 30.8 / 40.3 us on eight waves against 32.0 / 42.0 on four (CSWin stage 4 qkv / fc1); fp32 rows at K = 384 need eight waves
    // (a 32-row tile is 12 float4 per thread on four: no registers left for them next to 144 of weights)
    const int nw = (g.K == 512 || (a32 && g.K == 384)) ? 8 : 4;
    int nt;
    if (g.K == 256) nt = 4; else if (g.K == 384) nt = 3; else if (g.K == 512) nt = 2; else return MI355_EUNSUPPORTED;
    const int slabw = nw * nt * 16;
    if (g.N % slabw) return MI355_EUNSUPPORTED;
    const int nslab = g.N / slabw;
    const int slots = resident_slots(1) * (8 / nw);
    const int nstream = slots / nslab;
    const long ntile = ((long)g.M + 31) / 32;
    if (nstream < 1 || ntile < 4L * nstream) return MI355_EUNSUPPORTED;      // too few rows to amortise the resident weights
    *p = WslabPlan{nw, nt, nslab, nstream, slots, a32};
    return MI355_OK;
}
}  // namespace

int gemm16_wslab_check(const G16Args& g, int precision) {
    WslabPlan p;
    return wslab_plan(g, 1, precision, &p);
}

int gemm16_wslab(const G16Args& g, int out16, int precision, hipStream_t st) {
    WslabPlan p;
    if (wslab_plan(g, out16, precision, &p) != MI355_OK) return MI355_EUNSUPPORTED;
    const int nw = p.nw, nslab = p.nslab, nstream = p.nstream, slots = p.slots;
    const bool a32 = p.a32;
    MI355_TRACE(st, "gemm16_wslab_kernel<%s,K%d,%dw%s> M=%d N=%d%s", precision == MI355_PREC_FP16 ? "f16" : "bf16", g.K, nw, a32 ? ",x32" : "", g.M,
                g.N, g.act == MI355_ACT_GELU ? " gelu" : "");
#define GO4(T_, K_, NT_, NW_, A32_)                                                                               \
    do {                                                                                                          \
        if (g.act == MI355_ACT_GELU) gemm16_wslab_kernel<T_, K_, NT_, NW_, true, A32_><<<slots, NW_ * 64, 0, st>>>(g, nslab, nstream);  \
        else                         gemm16_wslab_kernel<T_, K_, NT_, NW_, false, A32_><<<slots, NW_ * 64, 0, st>>>(g, nslab, nstream); \
    } while (0)
#define GO2(T_)                                                                                                   \
    do {                                                                                                          \
        if (a32) { if (g.K == 256) GO4(T_, 256, 4, 4, true); else if (g.K == 384) GO4(T_, 384, 3, 8, true); else GO4(T_, 512, 2, 8, true); }      \
        else     { if (g.K == 256) GO4(T_, 256, 4, 4, false); else if (g.K == 384) GO4(T_, 384, 3, 4, false); else GO4(T_, 512, 2, 8, false); }  \
    } while (0)
    if (precision == MI355_PREC_FP16) GO2(_Float16); else GO2(__bf16);
#undef GO2
#undef GO4
    return MI355_OK;
}

}  // namespace mi355
