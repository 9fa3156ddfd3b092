// gemm16_p8.hip -- persistent 256 x 256 x 64 GEMM on the 16-bit engine: ONE workgroup of 8 waves per CU walks a list of output tiles,
// LDS-DMA runs seven half-tiles ahead of the matrix pipe and never drains (counted vmcnt, raw s_barrier), so a tile's prologue
// latency hides under the previous tile's epilogue and main loop.
//
//   Y (M x N) = resid + gamma * act( X16 (M x K) . W16^T (N x K) + bias )            same contract and bit-identical results as gemm16.hip
//
// Geometry (cdna_hip_programming.md "256^2 8-phase template" as the starting point):
//   waves 2 (M) x 4 (N), 128 x 64 outputs each = 8 x 4 MFMA 16x16x32 tiles, 64 MFMAs per wave and K-tile in four 16-MFMA quadrants;
//   LDS 128 KB = 2 K-tile buffers x 4 half-tile slots of 16 KB (128 rows x 128 B, source-side XOR swizzle as in gemm16.hip):
//       slot A0 = rows [0,64) of BOTH row groups, slot A1 = rows [64,128) of both, slot B0 / B1 = columns [0,32) / [32,64) of the
//       four column groups -- i.e. the halves are cut by WHEN they are consumed: phase 0 reads A0 + B0, phase 1 B1, phase 2 A1,
//       phase 3 nothing (B0 is still in registers), and a slot is refilled one phase after its last read;
//   phase = [ds_read fragments + issue ONE half-tile of LDS-DMA (2 x 1 KB per wave)] s_barrier [16 MFMAs] s_barrier; the two waves
//       of a SIMD (row groups wr = 0 / 1) run this program shifted by one barrier, so in every barrier interval one of them feeds
//       the matrix pipe while the other reads LDS and issues DMA;
//   DMA order (K-tile U): A0, B0, B1, A1; at phase p of K-tile T the half-tile 4T + p + 7 of the stream is issued:
//       p = 0: (T+1, A1)   p = 1: (T+2, A0)   p = 2: (T+2, B0)   p = 3: (T+2, B1)   then s_waitcnt vmcnt(6): everything up to
//       (T+1, A1) has landed, three half-tiles stay in flight across the barriers.  A rows are staged by the row group that reads
//       them (WAR / RAW against the shifted partner group cannot occur), B columns by all eight waves (hazard distances in DESIGN.md).
//   The stream of K-tiles is CONTINUOUS over the workgroup's output tiles: after the last K-tile of a tile the accumulators go out
//   (direct stores from the transposed MFMA tiles, bias / GELU / LayerScale / residual fused) while the next tile's first seven
//   half-tiles are already in flight.  Tiles are dealt XCD-contiguously: the 32 workgroups of an XCD work on neighbouring tiles
//   (shared A rows, the whole of W) out of one L2.
#include "gemm16.h"

namespace {
using namespace g16;

template <typename T, bool OUT16>
__global__ __launch_bounds__(512) void gemm16_p8_kernel(const G16Args g, const int tiles_m, const int tiles_n) {
    using v8 = typename Vec8<T>::t;
    using v4 = typename Vec8<T>::t4;
    constexpr int SLOT = 128 * BK;                              // elements per half-tile slot (16 KB)
    constexpr int BUF = 4 * SLOT;                               // elements per K-tile buffer (64 KB)
    constexpr int S_A0 = 0, S_B0 = SLOT, S_B1 = 2 * SLOT, S_A1 = 3 * SLOT;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * BUF * 2];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);
    const int nk = g.K / BK;

    // ---- this workgroup's tile list: XCD x owns the contiguous range [x0, x0 + xn) of the n-fastest tile order; its workgroups
    //      (blockIdx = x, x + 8, ...) take every per_xcd-th tile of that range ----------------------------------------------------
    const int ntiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int per_xcd = (gridDim.x - xcd + 7) >> 3;             // workgroups of this launch that sit on XCD `xcd`
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int x0 = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int xn = tq + (xcd < trm ? 1 : 0);
    const int my_first = x0 + slot_in_xcd;
    const int my_count = slot_in_xcd < xn ? (xn - slot_in_xcd + per_xcd - 1) / per_xcd : 0;
    if (my_count == 0) return;
    const int total_kt = my_count * nk;                         // length of this workgroup's K-tile stream

    // ---- DMA sources.  One instruction = 8 rows x 128 B; lane -> (row = lane >> 3, physical chunk = lane & 7) holding logical
    //      chunk (lane & 7) ^ row.  A half-tiles: this wave stages rows wc*16 + i*8 + lrow of ITS OWN row group's 64-row half;
    //      B half-tiles: rows w*16 + i*8 + lrow of the 128-row slot, slot row r <-> column (r / 32) * 64 + half * 32 + r % 32 ------
    const int lrow = lane >> 3, csw = ((lane & 7) ^ lrow) * 8;
    struct Cursor { const T* a[2]; const T* b[2]; int kt; int tile; };    // pointers of half 0 at k = 0; half 1 = + 64 rows / + 32 columns
    auto seek = [&](Cursor& c, int tile_no) {                  // tile_no: index into this workgroup's list
        const int tile = my_first + tile_no * per_xcd;
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ma = m0 + wr * 128 + wc * 16 + i * 8 + lrow;
            const int rb = wave * 16 + i * 8 + lrow;
            int nb = n0 + (rb >> 5) * 64 + (rb & 31);
            // tail tiles: rows / columns beyond the matrix are clamped (computed and discarded); the +64 / +32 halves clamp too
            c.a[i] = A + (long)(ma < g.M ? ma : g.M - 1) * g.lda + csw;
            c.b[i] = B + (long)(nb < g.N ? nb : g.N - 1) * g.ldb + csw;
        }
        c.kt = 0; c.tile = tile_no;
    };
    // offsets of the second halves, clamped per lane against the matrix edge (element offsets relative to the half-0 pointers)
    auto half1_a = [&](const Cursor& c, int i) -> const T* {
        const int tile = my_first + c.tile * per_xcd;
        const int ma = (tile / tiles_n) * 256 + wr * 128 + 64 + wc * 16 + i * 8 + lrow;
        return A + (long)(ma < g.M ? ma : g.M - 1) * g.lda + csw;
    };
    auto half1_b = [&](const Cursor& c, int i) -> const T* {
        const int tile = my_first + c.tile * per_xcd;
        const int rb = wave * 16 + i * 8 + lrow;
        const int nb = (tile % tiles_n) * 256 + (rb >> 5) * 64 + 32 + (rb & 31);
        return B + (long)(nb < g.N ? nb : g.N - 1) * g.ldb + csw;
    };
    const bool edge = (g.M & 255) || (g.N & 255);               // uniform: interior-only launches use the cheap +64 rows / +32 columns form

    auto dma = [&](const T* src, T* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // stage one half-tile of stream K-tile `u` (cursor c): which = 0 A0, 1 B0, 2 B1, 3 A1
    auto stage = [&](const Cursor& c, int u, int which) {
        T* base = lds + (u & 1) * BUF;
        const long k0 = (long)c.kt * BK;
        if (which == 0 || which == 3) {
            T* dst = base + (which == 0 ? S_A0 : S_A1) + (wr * 64 + wc * 16) * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const T* src = which == 0 ? c.a[i] : (edge ? half1_a(c, i) : c.a[i] + (long)64 * g.lda);
                dma(src + k0, dst + i * 8 * BK);
            }
        } else {
            T* dst = base + (which == 1 ? S_B0 : S_B1) + (wave * 16) * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const T* src = which == 1 ? c.b[i] : (edge ? half1_b(c, i) : c.b[i] + (long)32 * g.ldb);
                dma(src + k0, dst + i * 8 * BK);
            }
        }
    };
    auto advance = [&](Cursor& c) {                             // next K-tile of the stream
        if (++c.kt == nk) {
            if (c.tile + 1 < my_count) seek(c, c.tile + 1);
            else { c.kt = 0; c.tile = my_count; }               // past the end: never staged (callers test the stream index)
        }
    };

    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    v8 fa[4][2], fb[4][2];
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    const int off0 = ((fq ^ fsw) * 8), off1 = (((4 + fq) ^ fsw) * 8);
    auto read_a = [&](int u, int half) {                        // slot rows wr*64 + i*16 + frow
        const T* p = lds + (u & 1) * BUF + (half ? S_A1 : S_A0) + (wr * 64 + frow) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *reinterpret_cast<const v8*>(p + i * 16 * BK + off0);
            fa[i][1] = *reinterpret_cast<const v8*>(p + i * 16 * BK + off1);
        }
    };
    auto read_b = [&](int u, int half) {                        // slot rows wc*32 + j*16 + frow -> fb[half*2 + j]
        const T* p = lds + (u & 1) * BUF + (half ? S_B1 : S_B0) + (wc * 32 + frow) * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[half * 2 + j][0] = *reinterpret_cast<const v8*>(p + j * 16 * BK + off0);
            fb[half * 2 + j][1] = *reinterpret_cast<const v8*>(p + j * 16 * BK + off1);
        }
    };
#define P8_MMA(MH, NH)                                                                                          \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
                    acc[(MH) * 4 + i][(NH) * 2 + j] = mma16<T>(fb[(NH) * 2 + j][kk], fa[i][kk], acc[(MH) * 4 + i][(NH) * 2 + j]); \
        __builtin_amdgcn_s_setprio(0);                                                                          \
    } while (0)
#define P8_BAR() __builtin_amdgcn_s_barrier()

    // ---- prologue: stream half-tiles 0 .. 6 -------------------------------------------------------------------------------------
    Cursor c2;                                                   // cursor of stream K-tile T + 2 (T = the K-tile being computed)
    const T* a1p[2];                                             // A1 pointers (k included) of stream K-tile T + 1
    seek(c2, 0);
    stage(c2, 0, 0); stage(c2, 0, 1); stage(c2, 0, 2); stage(c2, 0, 3);
    advance(c2);                                                 // -> K-tile 1
    if (total_kt > 1) {
        stage(c2, 1, 0); stage(c2, 1, 1); stage(c2, 1, 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) a1p[i] = (edge ? half1_a(c2, i) : c2.a[i] + (long)64 * g.lda) + (long)c2.kt * BK;
        advance(c2);                                             // -> K-tile 2
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // K-tile 0 landed, three half-tiles of K-tile 1 still fly
    } else {
        a1p[0] = a1p[1] = A;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8_BAR();
    if (wr == 1) P8_BAR();                                       // one-interval shift of the second row group

    int kt_in_tile = 0, out_tile = 0;
    for (int T_ = 0; T_ < total_kt; ++T_) {
        const bool has1 = T_ + 1 < total_kt, has2 = T_ + 2 < total_kt;
        // ---- phase 0: A0 + B0 fragments; DMA (T+1, A1) ---------------------------------------------------------------------------
        read_b(T_, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(T_, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) {
            T* dst = lds + ((T_ + 1) & 1) * BUF + S_A1 + (wr * 64 + wc * 16) * BK;
            dma(a1p[0], dst);
            dma(a1p[1], dst + 8 * BK);
        }
        P8_BAR();
        P8_MMA(0, 0);
        P8_BAR();
        // ---- phase 1: B1 fragments; DMA (T+2, A0) ---------------------------------------------------------------------------------
        read_b(T_, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (has2) stage(c2, T_ + 2, 0);
        P8_BAR();
        P8_MMA(0, 1);
        P8_BAR();
        // ---- phase 2: A1 fragments; DMA (T+2, B0) ---------------------------------------------------------------------------------
        read_a(T_, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (has2) stage(c2, T_ + 2, 1);
        P8_BAR();
        P8_MMA(1, 1);
        P8_BAR();
        // ---- phase 3: no reads; DMA (T+2, B1); retire K-tile T+1 -----------------------------------------------------------------
        if (has2) {
            stage(c2, T_ + 2, 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) a1p[i] = (edge ? half1_a(c2, i) : c2.a[i] + (long)64 * g.lda) + (long)c2.kt * BK;
            advance(c2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        P8_BAR();
        P8_MMA(1, 0);
        P8_BAR();

        if (++kt_in_tile == nk) {
            // ---- epilogue of this output tile (transposed MFMA tiles: lane = one row, 4 consecutive columns) -------------------------
            const int tile = my_first + out_tile * per_xcd;
            const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
            // The tile leaves through a wave-private 2 KB LDS slab -- the part of slot A1 of the buffer just consumed that THIS wave
            // refills (rows wr*64 + wc*16 + [0,16)): nobody else touches it, its next DMA is issued by this wave after the epilogue in
            // program order, so no barrier is needed.  bias / GELU / LayerScale are applied in the accumulator layout (lane = one
            // row, 4 consecutive columns), the residual and the stores in the transposed one: 8 lanes per 128-byte row line, whole
            // lines per instruction (the direct 8 / 16-byte stores of gemm16.hip leave 32 / 64-byte runs, which a lone workgroup per
            // CU cannot hide behind neighbours: 40-50 us per tile on the fp32 + residual outputs before this).
            float* Cf = static_cast<float*>(g.C);
            T* Ch = static_cast<T*>(g.C);
            const int l15 = lane & 15, fq4 = lane >> 4;
            unsigned char* slab = lds_raw + ((size_t)((T_ & 1) * BUF + S_A1 + (wr * 64 + wc * 16) * BK)) * 2;
            const int srow = lane >> 3, sch = lane & 7;                  // read side: row srow (+8), 16-byte chunk sch
            const int ncol0 = n0 + wc * 64;
            f4 bias4[4], gam4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol0 + j * 16 + fq4 * 4;
                bias4[j] = (g.bias && n < g.N) ? *reinterpret_cast<const f4*>(g.bias + n) : f4{0.f, 0.f, 0.f, 0.f};
                gam4[j] = (g.gamma && n < g.N) ? *reinterpret_cast<const f4*>(g.gamma + n) : f4{1.f, 1.f, 1.f, 1.f};
            }
            if constexpr (OUT16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f4 v = acc[i][j] + bias4[j];
                        if (g.act == MI355_ACT_GELU) v = f4{gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w)};
                        if (g.gamma) v = v * gam4[j];
                        *reinterpret_cast<v4*>(slab + l15 * 128 + (((j * 2 + (fq4 >> 1)) ^ (l15 & 7)) * 16) + (fq4 & 1) * 8) =
                            v4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = h * 8 + srow;
                        const int m = m0 + wr * 128 + i * 16 + r, n = ncol0 + sch * 8;
                        const v8 o = *reinterpret_cast<const v8*>(slab + r * 128 + ((sch ^ (r & 7)) * 16));
                        if (m < g.M && n < g.N)                         // N % 8 == 0, no residual on this path (launcher)
                            *reinterpret_cast<v8*>(Ch + (long)m * g.ldc + n) = o;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                // fp32: the slab holds 16 rows x 32 columns per step (two of the four column tiles)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        f4 rr[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {                   // residual of this step: in flight while the slab is written
                            const int m = m0 + wr * 128 + i * 16 + h * 8 + srow, n = ncol0 + jh * 32 + sch * 4;
                            rr[h] = (g.resid && m < g.M && n < g.N) ? *reinterpret_cast<const f4*>(g.resid + (long)m * g.ldc + n)
                                                                    : f4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = jh * 2 + jj;
                            f4 v = acc[i][j] + bias4[j];
                            if (g.act == MI355_ACT_GELU) v = f4{gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w)};
                            if (g.gamma) v = v * gam4[j];
                            *reinterpret_cast<f4*>(slab + l15 * 128 + (((jj * 4 + fq4) ^ (l15 & 7)) * 16)) = v;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int r = h * 8 + srow;
                            const int m = m0 + wr * 128 + i * 16 + r, n = ncol0 + jh * 32 + sch * 4;
                            const f4 o = *reinterpret_cast<const f4*>(slab + r * 128 + ((sch ^ (r & 7)) * 16));
                            if (m < g.M && n < g.N) *reinterpret_cast<f4*>(Cf + (long)m * g.ldc + n) = o + rr[h];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
            kt_in_tile = 0;
            ++out_tile;
        }
    }
    if (wr == 0) P8_BAR();                                       // balance the shift: every wave executed the same number of barriers
#undef P8_MMA
#undef P8_BAR
}

}  // namespace

namespace mi355 {

// Launch the persistent kernel when the shape suits it: called by mi355_linear16_fwd (gemm16.hip).  Returns MI355_EUNSUPPORTED
// without touching anything when it does not apply.
int gemm16_p8(const g16::G16Args& g, int out16, int precision, hipStream_t st) {
    if ((g.K % g16::BK) || (g.N & 7) || (out16 && g.resid)) return MI355_EUNSUPPORTED;   // 16-bit out + residual: rounding point differs
    const int tiles_m = cdiv(g.M, 256), tiles_n = cdiv(g.N, 256);
    const long ntiles = (long)tiles_m * tiles_n;
    if (ntiles > (1L << 30)) return MI355_EUNSUPPORTED;
    int grid = resident_slots(1);
    if (grid > ntiles) grid = (int)ntiles;
    if (precision == MI355_PREC_FP16) {
        if (out16) gemm16_p8_kernel<_Float16, true><<<grid, 512, 0, st>>>(g, tiles_m, tiles_n);
        else       gemm16_p8_kernel<_Float16, false><<<grid, 512, 0, st>>>(g, tiles_m, tiles_n);
    } else {
        if (out16) gemm16_p8_kernel<__bf16, true><<<grid, 512, 0, st>>>(g, tiles_m, tiles_n);
        else       gemm16_p8_kernel<__bf16, false><<<grid, 512, 0, st>>>(g, tiles_m, tiles_n);
    }
    return MI355_OK;
}

}  // namespace mi355
