// gemm16_wreg.hip -- Y (M x N, fp32) = resid + X16 (M x K) . W16^T (N x K) + bias  for SQUARE, SHORT products (N = K = 256 / 384) on gfx950:
// the projection behind an attention core at token widths 256 ... 384 (XCiT XCA proj, xcit.py:263; CSWin stage 3 proj, cswin.py:192).
//
// Why a third GEMM schedule: at N = K = 384 the product is 14.8 GFLOP over 192 MB (16-bit X in, fp32 residual in, fp32 Y out) -- 6 us of
// matrix pipe against 40 us of HBM at the copy rate of these boxes.  The tile kernels of the engine stream BOTH operands through LDS
// per output tile and pay an epilogue per tile; on this shape they reach 2.8 TB/s (70 us: profiles/r05_XCABlock_kernel_seq.txt, and the
// 256 x 256 tile leaves half of its second column tile empty at N = 384).  Here the WEIGHTS ARE STATIONARY IN REGISTERS:
//
//   workgroup  = 8 waves, persistent (one per CU); wave w owns output columns [w * N/8, (w + 1) * N/8) for EVERY row: its N/8 x K slice of W
//                lives in VGPRs as MFMA A-operand fragments for the whole kernel (N = K = 384: 3 column tiles x 12 k-steps x 4 = 144 VGPRs);
//   row tile   = 32 rows of X, staged once in LDS (double-buffered: the next tile's rows are in flight during the MFMAs) and read by all
//                eight waves as B-operand fragments; Y^T tiles = W . X^T, so a lane holds 4 consecutive output columns of one row and the
//                residual load / the store are 16-byte pieces of 192-byte (N = 384) row segments;
//   per tile   : residual loads issued first, 2 x NT x K/32 MFMAs per wave, ONE workgroup barrier.
// HBM traffic = X once + residual once + Y once; W is read once per workgroup (75 MB over the chip at N = K = 384, L2-resident).
// A row's K steps are added in ascending order on the same MFMA instruction and the epilogue is (acc + bias) + resid: bit-identical to
// gemm16_p8 / gemm16_pa on these shapes (tests/test_round6_kernels_gpu.py).
#include "gemm16.h"
#include "bufops.h"
#include <type_traits>

namespace {

using namespace g16;

template <typename T, int K, int NT, bool RESID, int STATS = 0>       // STATS: 0 none, 1 row statistics, 2 LayerNorm'ed 16-bit copy
__global__ __launch_bounds__(512, 1) void gemm16_wreg_kernel(const G16Args g) {
    using v8 = typename Vec8<T>::t;
    constexpr int NW = 8, NTHR = 512, RT = 2, ROWS = RT * 16;
    constexpr int KS = K / 32;
    constexpr int NCW = NT * 16;                  // columns per wave
    constexpr int N = NW * NCW;
    constexpr int XP = K + 8;                     // LDS row pitch (elements)
    constexpr int CH = ROWS * (K / 8);            // 16-byte chunks of a row tile
    constexpr int NLD = (CH + NTHR - 1) / NTHR;   // chunks per thread
    __shared__ __attribute__((aligned(16))) unsigned short s_x[2][ROWS * XP];
    __shared__ float s_part[STATS ? 2 : 1][2][NW][ROWS];   // STATS: per-wave partial (mean, M2) of the tile's rows, double-buffered by tile parity

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, gq = lane >> 4;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ W = static_cast<const T*>(g.B);
    float* __restrict__ Y = static_cast<float*>(g.C);
    const int n0 = wave * NCW;

    // ---- this wave's slice of W: A-operand fragments (row = output column n0 + nt*16 + l15, k = ks*32 + gq*8 + [0,8)) ----------------
    v8 wfr[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wfr[nt][ks] = *reinterpret_cast<const v8*>(W + (long)(n0 + nt * 16 + l15) * g.ldb + ks * 32 + gq * 8);
    f4 bias4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        bias4[nt] = g.bias ? *reinterpret_cast<const f4*>(g.bias + n0 + nt * 16 + gq * 4) : f4{0.f, 0.f, 0.f, 0.f};

    float rgmax = 0.f;
    const long ntile = ((long)g.M + ROWS - 1) / ROWS;
    // chunk c of a row tile: row c / (K/8), 8 elements at column (c % (K/8)) * 8
    int crow[NLD], ccol[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int c = t + j * NTHR;
        crow[j] = c / (K / 8);
        ccol[j] = (c % (K / 8)) * 8;
    }
    v8 areg[NLD];
    auto fetch = [&](long tile) {                 // rows beyond M come back as zeros from the buffer range check (their outputs are dropped)
        const long r0 = tile * ROWS;
        const long left = (long)g.M - r0;
        const int rows = (int)(left < ROWS ? left : ROWS);
        const rsrc_t rs = make_rsrc(A + r0 * g.lda, (bufops_u32)(((long)(rows - 1) * g.lda + K) * 2));
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const bool live = (CH % NTHR == 0) || (t + j * NTHR < CH);
            const bufops_u32 off = live ? (bufops_u32)((crow[j] * g.lda + ccol[j]) * 2) : OOB;
            areg[j] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if ((CH % NTHR == 0) || (t + j * NTHR < CH))
                *reinterpret_cast<v8*>(&s_x[buf][crow[j] * XP + ccol[j]]) = areg[j];
    };

    long tile = blockIdx.x;
    if (tile < ntile) {
        fetch(tile);
        commit(0);
    }
    __syncthreads();
    int buf = 0, tile_par = 0;
    for (; tile < ntile; tile += gridDim.x) {
        const long r0 = tile * ROWS;
        const long left = (long)g.M - r0;
        const int rows = (int)(left < ROWS ? left : ROWS);
        const long next = tile + gridDim.x;
        // ---- everything this tile needs from HBM is requested up front: its residual rows, then the next tile's X rows ----------------
        // per-tile descriptors (wave-uniform): a lane's offset is (row * ldc + column) * 4, rows beyond M fall outside num_records
        const bufops_u32 ybytes = (bufops_u32)(((long)(rows - 1) * g.ldc + N) * 4);
        f4 rr[RT][NT];
        if constexpr (RESID) {
            const rsrc_t rres = make_rsrc(g.resid + r0 * g.ldc, ybytes);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bufops_u32 off = (bufops_u32)(((rt * 16 + l15) * g.ldc + n0 + nt * 16 + gq * 4) * 4);
                    rr[rt][nt] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 0));
                }
        }
        if (next < ntile) fetch(next);
        // ---- Y^T tiles = W . X^T: lane (l15, gq) holds columns n0 + nt*16 + gq*4 + [0,4) of row rt*16 + l15 ------------------------------
        f4 acc[RT][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const v8 xf = *reinterpret_cast<const v8*>(&s_x[buf][(rt * 16 + l15) * XP + ks * 32 + gq * 8]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = mma16<T>(wfr[nt][ks], xf, acc[rt][nt]);
            }
        }
        if (next < ntile) commit(buf ^ 1);
        // ---- epilogue: (acc + bias) + resid, 16-byte stores -----------------------------------------------------------------------------
        const rsrc_t ry = make_rsrc(Y + r0 * g.ldc, ybytes);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f4 v = acc[rt][nt] + bias4[nt];
                if constexpr (RESID) v = v + rr[rt][nt];
                if constexpr (STATS != 0) acc[rt][nt] = v;
                const bufops_u32 off = (bufops_u32)(((rt * 16 + l15) * g.ldc + n0 + nt * 16 + gq * 4) * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), ry, off, 0, 0);
            }
        }
        if constexpr (STATS != 0) {
            // LayerNorm statistics of the rows just written WITHOUT a second workgroup barrier: every lane forms (mean, M2 = sum of squared
            // deviations) of its own NCW / 4 values exactly (two passes over registers), partials are merged pairwise with the parallel-variance
            // update  mean = mean_a + d n_b / n,  M2 = M2_a + M2_b + d^2 n_a n_b / n,  d = mean_b - mean_a  -- across the four quarter-row
            // lane groups by two xor-shuffles, across the eight waves through LDS in wave order (fixed order: reproducible, and as robust as
            // the two-pass form against |mean| >> std).  The exchange area is double-buffered by tile parity, so the tile's ONE barrier
            // (the one the X double buffer needs anyway) also publishes the partials.
            constexpr float NL = (float)(NT * 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) sum += (acc[rt][nt].x + acc[rt][nt].y) + (acc[rt][nt].z + acc[rt][nt].w);
                float m = sum * (1.0f / NL), q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f4 d = acc[rt][nt] - m;
                    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
                float n = NL;
#pragma unroll
                for (int sh = 16; sh <= 32; sh <<= 1) {               // equal counts on both sides: mean = (a + b) / 2, M2 += d^2 n / 2
                    const float mo = __shfl_xor(m, sh, WAVE), qo = __shfl_xor(q, sh, WAVE);
                    const float d = mo - m;
                    q = (q + qo) + d * d * (n * 0.5f);
                    m = 0.5f * (m + mo);                               // symmetric in the two partners: both lanes get the same bits
                    n *= 2.0f;
                }
                if (gq == 0) { s_part[tile_par][0][wave][rt * 16 + l15] = m; s_part[tile_par][1][wave][rt * 16 + l15] = q; }
            }
        }
        __syncthreads();                              // next tile's rows complete in s_x[buf ^ 1]; everybody is done reading s_x[buf]; partials published
        if constexpr (STATS != 0) {
            constexpr float NWV = (float)NCW;          // values behind one wave's partial
            const float invN = 1.0f / (float)N;
            float mean[RT], rstd[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float m = s_part[tile_par][0][0][rt * 16 + l15], q = s_part[tile_par][1][0][rt * 16 + l15], n = NWV;
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const float mb = s_part[tile_par][0][w][rt * 16 + l15], qb = s_part[tile_par][1][w][rt * 16 + l15];
                    const float d = mb - m, nn = n + NWV;
                    q = (q + qb) + d * d * (n * NWV / nn);
                    m = m + d * (NWV / nn);
                    n = nn;
                }
                mean[rt] = m;
                rstd[rt] = 1.0f / sqrtf(q * invN + g.ln_eps);
            }
            if constexpr (STATS == 1) {
                if (gq == 0) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (wave == rt && rt * 16 + l15 < rows) {        // wave rt writes the 16 rows of row tile rt: one 8-byte store per row
                            float2 st;
                            st.x = mean[rt]; st.y = rstd[rt];
                            *reinterpret_cast<float2*>(g.row_stats + (r0 + rt * 16 + l15) * 2) = st;
                        }
                    }
                }
            } else {
                // the next LayerNorm, applied to the values this lane still holds: (v - mean) * rstd * w + b -> 16 bit, 8-byte stores
                typedef T t4 __attribute__((ext_vector_type(4)));
                const rsrc_t ru = make_rsrc(static_cast<T*>(g.ln16_out) + r0 * g.ln16_ld, (bufops_u32)(((long)(rows - 1) * g.ln16_ld + N) * 2));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        // the affine vectors are re-read per tile (L2 hits): kept for the kernel's lifetime they would cost 2 NT more VGPR quads
                        const f4 lw = *reinterpret_cast<const f4*>(g.ln16_w + n0 + nt * 16 + gq * 4);
                        const f4 lb = *reinterpret_cast<const f4*>(g.ln16_b + n0 + nt * 16 + gq * 4);
                        const f4 o = (acc[rt][nt] - mean[rt]) * rstd[rt] * lw + lb;
                        if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4_f(rgmax, o);
                        const t4 h = t4{(T)o.x, (T)o.y, (T)o.z, (T)o.w};
                        const bufops_u32 off = (bufops_u32)(((rt * 16 + l15) * g.ln16_ld + n0 + nt * 16 + gq * 4) * 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, h), ru, off, 0, 0);
                    }
                }
            }
            tile_par ^= 1;
        }
        buf ^= 1;
    }
    if constexpr (STATS == 2 && std::is_same<T, _Float16>::value) rg_report_f(rgmax, g.ovf, 2u);
}

}  // namespace

namespace mi355 {

// MI355_EUNSUPPORTED (nothing launched) unless the shape is one of the square short products this schedule is built for.
int gemm16_wreg(const G16Args& g, int out16, int precision, hipStream_t st) {
    if (out16 || g.act != MI355_ACT_NONE || g.gamma || g.resid_period || g.lnc_a || g.rowtau) return MI355_EUNSUPPORTED;
    if (g.row_stats && (!g.resid || (reinterpret_cast<uintptr_t>(g.row_stats) & 7))) return MI355_EUNSUPPORTED;
    if (g.ln16_out && g.K != 256) return MI355_EUNSUPPORTED;      // the LayerNorm-emitting epilogue is built at N = K = 256 (at 384 it would spill: 256 VGPRs + 20 B)
    if (g.ln16_out && (!g.resid || g.row_stats || !g.ln16_w || !g.ln16_b || !aligned16(g.ln16_w) || !aligned16(g.ln16_b) || !aligned16(g.ln16_out) ||
                       (g.ln16_ld & 3) || g.ln16_ld < g.N))
        return MI355_EUNSUPPORTED;
    if (!(g.N == g.K && (g.K == 256 || g.K == 384))) return MI355_EUNSUPPORTED;
    if (g.M < 32 || (g.lda & 7) || (g.ldb & 7) || (g.ldc & 3) || g.ldb < g.K) return MI355_EUNSUPPORTED;
    if ((long)32 * g.ldc * 4 >= (1L << 31) || (long)32 * g.lda * 2 >= (1L << 31)) return MI355_EUNSUPPORTED;
    if (precision != MI355_PREC_FP16 && precision != MI355_PREC_BF16) return MI355_EUNSUPPORTED;
    const long ntile = ((long)g.M + 31) / 32;
    const int ncu = resident_slots(1);
    const int grid = (int)(ntile < ncu ? ntile : ncu);
    MI355_TRACE(st, "gemm16_wreg_kernel<%s,%s> M=%d N=%d K=%d", precision == MI355_PREC_FP16 ? "f16" : "bf16", g.ln16_out ? "resid+ln16" : (g.row_stats ? "resid+stats" : (g.resid ? "resid" : "plain")), g.M, g.N, g.K);
#define GO(T_, K_, NT_)                                                                  \
    do {                                                                                 \
        if (g.ln16_out) { if constexpr (K_ == 256) gemm16_wreg_kernel<T_, K_, NT_, true, 2><<<grid, 512, 0, st>>>(g); }   \
        else if (g.row_stats) gemm16_wreg_kernel<T_, K_, NT_, true, 1><<<grid, 512, 0, st>>>(g);   \
        else if (g.resid) gemm16_wreg_kernel<T_, K_, NT_, true><<<grid, 512, 0, st>>>(g);     \
        else         gemm16_wreg_kernel<T_, K_, NT_, false><<<grid, 512, 0, st>>>(g);    \
    } while (0)
    if (precision == MI355_PREC_FP16) { if (g.K == 384) GO(_Float16, 384, 3); else GO(_Float16, 256, 2); }
    else                              { if (g.K == 384) GO(__bf16, 384, 3); else GO(__bf16, 256, 2); }
#undef GO
    return MI355_OK;
}

}  // namespace mi355
