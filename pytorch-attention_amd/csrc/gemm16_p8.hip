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
//   Last partial round: ntiles = F * ncu + R.  Every workgroup walks F whole tiles; each of the R left-over tiles is cut along K
//   into S = ncu / R chunks, one per workgroup (the stream of a workgroup simply ends with a shorter entry).  Chunks s > 0 dump
//   their accumulators in register layout (1 KB per wave-instruction) into a caller-provided fp32 slab, release at agent scope and
//   count themselves in on a per-tile word; chunk 0 polls that word (relaxed), acquires, adds the slabs and runs the normal
//   epilogue -- N = 768 outputs (591 tiles on 256 CUs = 2.31 rounds) take 2.33 rounds instead of 3.  Without a workspace S = 1.
#include <atomic>
#include <type_traits>
#include "gemm16.h"
#include "bufops.h"

namespace {
using namespace g16;
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

}  // namespace
struct P8Plan {
    int tiles_m, tiles_n;
    int full;              // F: whole tiles per workgroup
    int left, split;       // R left-over tiles, each cut into S K-chunks (S >= 1; the chunk units go to workgroups 0 .. R*S-1)
    float* part;           // (R, S-1, 8 waves, 8192) fp32 partial accumulators
    unsigned long long* arrive;   // (R, S-1, 8 waves) flags {tag, ~tag}: the wave of chunk s > 0 that has published its slab.  Never zeroed:
                           // the tag is unique per launch, so whatever an earlier launch (or other data) left there cannot match
    unsigned tag;
    unsigned* herr;        // pinned host failure word (api.hip sync_err_word) or null
    unsigned spin;         // poll budget of chunk 0 (sweeps of ~1 us)
};
namespace {

// FOLD (16-bit outputs only): LayerNorm fold, consumer side (ln_fold.hip) -- the operand rows are x - c, the weights carry the
// LayerNorm gain, and the row's rstd / mean correction arrive as rowtau[m] = {rho, tau}:  y = act(rho * acc + tau * colsum[n] + bias[n]).
template <typename T, bool OUT16, bool FOLD = false>
__global__ __launch_bounds__(512) void gemm16_p8_kernel(const G16Args g, const P8Plan pl) {
    static_assert(!FOLD || OUT16, "the folding epilogue exists for 16-bit outputs only");
    const int tiles_n = pl.tiles_n;
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
    //      Entries 0 .. F-1 of a workgroup's list are whole tiles out of the first F * gridDim tiles; entry F (if any) is its K-chunk
    //      of a left-over tile ----------------------------------------------------------------------------------------------------
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int gq = gridDim.x >> 3, gr = gridDim.x & 7;
    const int per_xcd = gq + (xcd < gr ? 1 : 0);                // workgroups of this launch that sit on XCD `xcd`
    const int my_first = pl.full * (xcd * gq + (xcd < gr ? xcd : gr)) + slot_in_xcd;      // XCD x owns F * per_xcd consecutive tiles
    const int nfull = pl.full;
    const bool has_chunk = (int)blockIdx.x < pl.left * pl.split;
    const int ch_tile_r = has_chunk ? (int)blockIdx.x / pl.split : 0, ch_s = has_chunk ? (int)blockIdx.x % pl.split : 0;
    const int ch_tile = pl.full * (int)gridDim.x + ch_tile_r;
    const int ch_k0 = has_chunk ? (int)((long)ch_s * nk / pl.split) : 0, ch_k1 = has_chunk ? (int)((long)(ch_s + 1) * nk / pl.split) : 0;
    const int my_count = nfull + (has_chunk ? 1 : 0);
    if (my_count == 0) return;
    const int total_kt = nfull * nk + (ch_k1 - ch_k0);          // length of this workgroup's K-tile stream
    auto entry_tile = [&](int e) { return e < nfull ? my_first + e * per_xcd : ch_tile; };

    // ---- DMA sources.  One instruction = 8 rows x 128 B; lane -> (row = lane >> 3, physical chunk = lane & 7) holding logical
    //      chunk (lane & 7) ^ row.  A half-tiles: this wave stages rows wc*16 + i*8 + lrow of ITS OWN row group's 64-row half;
    //      B half-tiles: rows w*16 + i*8 + lrow of the 128-row slot, slot row r <-> column (r / 32) * 64 + half * 32 + r % 32 ------
    const int lrow = lane >> 3, csw = ((lane & 7) ^ lrow) * 8;
    struct Cursor { const T* a[2]; const T* b[2]; int kt; int kend; int tile; };   // pointers of half 0 at k = 0; half 1 = + 64 rows / + 32 columns
    auto seek = [&](Cursor& c, int tile_no) {                  // tile_no: index into this workgroup's list
        const int tile = entry_tile(tile_no);
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
        c.kt = tile_no < nfull ? 0 : ch_k0;
        c.kend = tile_no < nfull ? nk : ch_k1;
        c.tile = tile_no;
    };
    // offsets of the second halves, clamped per lane against the matrix edge (element offsets relative to the half-0 pointers)
    auto half1_a = [&](const Cursor& c, int i) -> const T* {
        const int tile = entry_tile(c.tile);
        const int ma = (tile / tiles_n) * 256 + wr * 128 + 64 + wc * 16 + i * 8 + lrow;
        return A + (long)(ma < g.M ? ma : g.M - 1) * g.lda + csw;
    };
    auto half1_b = [&](const Cursor& c, int i) -> const T* {
        const int tile = entry_tile(c.tile);
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
        if (++c.kt == c.kend) {
            if (c.tile + 1 < my_count) seek(c, c.tile + 1);
            else { c.kt = 0; c.kend = 1 << 30; c.tile = my_count; }   // past the end: never staged (callers test the stream index)
        }
    };

    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    float rgmax = 0.f;                                          // fp16 range guard over the 16-bit values this lane converts (common.h)
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

    const bool producer = has_chunk && ch_s > 0;                // the stream ends with a K-chunk whose partial sums go to the slab
    int kt_in_tile = 0, out_tile = 0;
    int entry_len = nfull > 0 ? nk : ch_k1 - ch_k0;
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

        if (++kt_in_tile == entry_len && !(producer && T_ + 1 == total_kt)) {
            // Both row groups run their epilogues AT THE SAME TIME: group 0 waits one barrier for the partner's last MFMA interval,
            // and group 1 re-creates the one-interval shift with an extra barrier before the next entry (left shifted, the partner's
            // epilogue could only start after this group's had finished: the next barrier it needs is this group's first of the next tile).
            if (wr == 0) P8_BAR();
            const bool is_chunk = out_tile >= nfull;
            if (is_chunk && pl.split > 1) {
                // ---- K-chunk 0: wait for the 8 * (S-1) partner waves, add their partial accumulators ------------------------------------
                // Bounded like every other inter-workgroup wait of the library: the partner chunks are workgroups of the SAME launch with
                // one workgroup per CU, so they run unless the device offers fewer CUs than the grid; a wait that runs out reports
                // through mi355_sync_status (code 4) instead of hanging the queue.
                // lane s < S-1 of owner wave w watches the flag of partner (chunk s + 1, wave w): those are exactly the slabs this wave adds
                const unsigned long long want = ((unsigned long long)(~pl.tag) << 32) | pl.tag;
                const unsigned long long* fl = pl.arrive + ((long)ch_tile_r * (pl.split - 1) + (lane < pl.split - 1 ? lane : 0)) * 8 + wave;
                unsigned spins = 0;
                for (;;) {
                    const unsigned long long got = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__ballot(got != want) == 0) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > pl.spin) {
                        if (lane == 0 && pl.herr) __hip_atomic_store(pl.herr, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
                // Ordering relied upon (MI355X_MICROARCH.md "Valid forms": sc1 stores AND sc1 loads on both sides + a drained queue in
                // front of the flag): the producer's slab stores are write-through and complete (vmcnt(0)) before its flag store is
                // issued; the slab loads below bypass this CU's L1 (sc1), so once the flag is seen they read what the producer wrote.
                // The barrier keeps the COMPILER from hoisting the buffer loads above the poll loop (relaxed atomics do not order them).
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                for (int s = 0; s < pl.split - 1; ++s) {                    // sc1 loads: served below this CU's L1, coherent with the sc1 stores
                    const float* slab = pl.part + (((long)ch_tile_r * (pl.split - 1) + s) * 8 + wave) * 8192;
                    const rsrc_t rs = make_rsrc(slab, 32768u);
#pragma unroll
                    for (int i0 = 0; i0 < 8; i0 += 4) {              // sixteen 1 KB loads in flight per step (the fragment registers are free here)
                        f4 pv[4][4];
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                pv[ii][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rs, (bufops_u32)lane * 16u, (bufops_u32)((i0 + ii) * 4 + j) * 1024u, AUX_SC1));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i0 + ii][j] = acc[i0 + ii][j] + pv[ii][j];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- epilogue of this output tile (transposed MFMA tiles: lane = one row, 4 consecutive columns) -------------------------
            const int tile = entry_tile(out_tile);
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
                // FOLD: a lane holds ONE row per row tile (l15) and four columns per column tile: 4 float4 of colsum per output tile and
                // one float2 of rowtau per row tile, fetched one row tile ahead (no LayerScale on this path: gam4 is dead)
                typedef float f2_ __attribute__((ext_vector_type(2)));
                f4 cs4[4];
                f2_ rt_next = f2_{1.f, 0.f};
                auto load_rt = [&](int i) -> f2_ {
                    const int m = m0 + wr * 128 + i * 16 + l15;
                    return m < g.M ? *reinterpret_cast<const f2_*>(g.rowtau + 2 * (long)m) : f2_{1.f, 0.f};
                };
                if constexpr (FOLD) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = ncol0 + j * 16 + fq4 * 4;
                        cs4[j] = n < g.N ? *reinterpret_cast<const f4*>(g.colsum + n) : f4{0.f, 0.f, 0.f, 0.f};
                    }
                    rt_next = load_rt(0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f2_ rt = rt_next;
                    if constexpr (FOLD) { if (i < 7) rt_next = load_rt(i + 1); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f4 v;
                        if constexpr (FOLD) v = acc[i][j] * rt.x + (cs4[j] * rt.y + bias4[j]);
                        else v = acc[i][j] + bias4[j];
                        if (g.act == MI355_ACT_GELU) v = gelu16_fast4(v);
                        if constexpr (!FOLD) { if (g.gamma) v = v * gam4[j]; }
                        if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4(rgmax, v);
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
                // periodic residual table: the wave's 128 rows are consecutive, so one modulo per tile and a conditional subtract per row
                const int per = g.resid_period, mr0 = per ? (m0 + wr * 128) % per : 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        f4 rr[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {                   // residual of this step: in flight while the slab is written
                            const int m = m0 + wr * 128 + i * 16 + h * 8 + srow, n = ncol0 + jh * 32 + sch * 4;
                            int mr = m;
                            if (per) {
                                mr = mr0 + i * 16 + h * 8 + srow;
                                mr -= mr >= per ? per : 0;
                            }
                            rr[h] = (g.resid && m < g.M && n < g.N) ? *reinterpret_cast<const f4*>(g.resid + (long)mr * g.ldc + n)
                                                                    : f4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = jh * 2 + jj;
                            f4 v = acc[i][j] + bias4[j];
                            if (g.act == MI355_ACT_GELU) v = gelu_fast4(v);
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
            entry_len = out_tile < nfull ? nk : ch_k1 - ch_k0;
            if (wr == 1 && out_tile < my_count) P8_BAR();
        }
    }
    if (producer) {
        if (wr == 0) P8_BAR();
        // ---- K-chunk s > 0 of a left-over tile: publish the partial accumulators, count this wave in, done ---------------------
        // (buffer addressing: one descriptor + one VGPR of lane offset + immediate offsets -- 32 flat addresses would spill)
        float* slab = pl.part + (((long)ch_tile_r * (pl.split - 1) + (ch_s - 1)) * 8 + wave) * 8192;
        const rsrc_t rs = make_rsrc(slab, 32768u);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, acc[i][j]), rs, (bufops_u32)lane * 16u,
                                                       (bufops_u32)(i * 4 + j) * 1024u, AUX_SC1);
        // write-through (sc1) stores + a drained queue ARE the publish (MI355X_MICROARCH.md "publish-large"): an agent-scope release
        // fence here would write back every dirty line of the XCD's L2 -- megabytes of output tiles -- once per wave (measured: the
        // split round then cost more than a whole tile)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(pl.arrive + ((long)ch_tile_r * (pl.split - 1) + (ch_s - 1)) * 8 + wave,
                               ((unsigned long long)(~pl.tag) << 32) | pl.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (OUT16 && std::is_same<T, _Float16>::value) rg_report(rgmax, g.ovf, 3u);
    // barrier balance: 1 + 8 * total_kt + my_count per wave (group 1: prologue shift + (my_count - 1) re-shifts; group 0: my_count re-alignments)
#undef P8_MMA
#undef P8_BAR
}

}  // namespace

namespace mi355 {

// Workspace of the split last round: partial accumulators of the K-chunks s > 0 of the left-over tiles + one 8-byte flag per
// publishing wave.
static void p8_plan(int M, int N, int K, int ncu, P8Plan& pl, int& grid, size_t& part_bytes, size_t& arrive_bytes) {
    pl.tiles_m = cdiv(M, 256); pl.tiles_n = cdiv(N, 256);
    const long ntiles = (long)pl.tiles_m * pl.tiles_n;
    grid = ntiles < ncu ? (int)ntiles : ncu;
    pl.full = (int)(ntiles / ncu);
    pl.left = (int)(ntiles - (long)pl.full * ncu);
    const int nk = K / g16::BK;
    int S = pl.left > 0 ? ncu / pl.left : 1;
    // a chunk saves nk * (1 - 1/S) K-tiles of ~1.75 us and costs a slab round trip (~15-20 us): only long reductions split
    if (nk < 24 || !opt_gemm_splitk()) S = 1;
    if (S > nk / 8) S = nk / 8;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
    pl.split = S;
    if (pl.full == 0) grid = pl.left * S;                      // fewer tiles than CUs: only the chunk units exist
    part_bytes = (size_t)pl.left * (S - 1) * 8 * 8192 * sizeof(float);
    arrive_bytes = ((size_t)pl.left * (S - 1) * 8 * sizeof(unsigned long long) + 255) & ~(size_t)255;
}

size_t gemm16_p8_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < g16::BK || (K % g16::BK)) return 0;
    P8Plan pl{};
    int grid;
    size_t pb, ab;
    p8_plan(M, N, K, resident_slots(1), pl, grid, pb, ab);
    return pl.split > 1 ? pb + ab : 0;
}

// Launch the persistent kernel when the shape suits it: called by mi355_linear16_fwd (gemm16.hip).  Returns MI355_EUNSUPPORTED
// without touching anything when it does not apply.  `ws` (may be null) enables the split last round.
int gemm16_p8(const g16::G16Args& g, int out16, int precision, void* ws, size_t ws_bytes, hipStream_t st) {
    if ((g.K % g16::BK) || (g.N & 7) || (out16 && g.resid)) return MI355_EUNSUPPORTED;
    if (g.resid_period && (g.resid_period < 128 || out16 || !g.resid)) return MI355_EUNSUPPORTED;   // 16-bit out + residual: rounding point differs
    if ((long)cdiv(g.M, 256) * cdiv(g.N, 256) > (1L << 30)) return MI355_EUNSUPPORTED;
    // LayerNorm fold, consumer side: its argument checks belong up here, before the trace scope opens (a refused call must not leave an
    // event pair under a kernel tag)
    if (g.rowtau && (!out16 || !g.colsum || g.gamma || !aligned16(g.colsum) || (reinterpret_cast<uintptr_t>(g.rowtau) & 7u))) return MI355_EUNSUPPORTED;
    P8Plan pl{};
    int grid;
    size_t pb, ab;
    p8_plan(g.M, g.N, g.K, resident_slots(1), pl, grid, pb, ab);
    // The split round synchronises workgroups through flags that carry a per-launch tag from a host-side counter.  A launch recorded
    // by hipGraph capture replays with the SAME arguments -- the same tag -- so on the second replay the flags of the first already
    // match and chunk 0 would add stale or half-written slabs: under capture the left-over tiles run whole (as the exchange kernels
    // of the channel-attention family step aside, api.hip stream_is_capturing).  The pinned error word is not allocated there either
    // (hipHostMalloc is illegal during capture; without a split nothing reports through it).
    const bool capturing = stream_is_capturing(st);
    if (pl.split > 1 && (capturing || ws == nullptr || ws_bytes < pb + ab || !aligned16(ws))) {      // no workspace: whole left-over tiles
        pl.split = 1;
        if (pl.full == 0) grid = pl.left;
    }
    pl.herr = capturing ? nullptr : sync_err_word();
    pl.spin = spin_limit();
    if (pl.split > 1) {
        if (int rc = sync_pending("gemm16_p8")) return rc;
        pl.arrive = static_cast<unsigned long long*>(ws);
        pl.part = reinterpret_cast<float*>(static_cast<char*>(ws) + ab);
        static std::atomic<unsigned> launch_tag{0x5EED0000u};
        do pl.tag = launch_tag.fetch_add(1u, std::memory_order_relaxed) + 1u; while (pl.tag == 0u);
    }
    MI355_TRACE(st, "gemm16_p8_kernel<%s,%s%s> M=%d N=%d K=%d%s", precision == MI355_PREC_FP16 ? "f16" : "bf16", out16 ? "out16" : "out32",
                g.rowtau ? ",fold" : "", g.M, g.N, g.K, g.act == MI355_ACT_GELU ? " gelu" : "");
    if (g.rowtau) {                                            // LayerNorm fold, consumer side (arguments checked above)
        if (precision == MI355_PREC_FP16) gemm16_p8_kernel<_Float16, true, true><<<grid, 512, 0, st>>>(g, pl);
        else                              gemm16_p8_kernel<__bf16, true, true><<<grid, 512, 0, st>>>(g, pl);
        return MI355_OK;
    }
    if (precision == MI355_PREC_FP16) {
        if (out16) gemm16_p8_kernel<_Float16, true><<<grid, 512, 0, st>>>(g, pl);
        else       gemm16_p8_kernel<_Float16, false><<<grid, 512, 0, st>>>(g, pl);
    } else {
        if (out16) gemm16_p8_kernel<__bf16, true><<<grid, 512, 0, st>>>(g, pl);
        else       gemm16_p8_kernel<__bf16, false><<<grid, 512, 0, st>>>(g, pl);
    }
    return MI355_OK;
}

}  // namespace mi355
