// double_attn_fused.hip -- A2-Net DoubleAttention forward (double_attention.py:32-48) in two passes over the image instead of the
// seven launches of double_attn.hip (which wrote and re-read the fp32 (A | B | V) tensor -- 1.5x the input -- three times).
//
//   pass 1  one workgroup per (image, pixel range), 8 waves, persistent over 32-pixel tiles of its range:
//             x tile (C x 32 px, fp32, NCHW rows) --LDS-DMA--> raw tile --4x4 register transposes--> X^T (32 px x C, 16 bit) in LDS
//             S^T = X^T [WA | WB | WV]^T + bias: wave w owns output channels 16w..16w+15 of EACH of the three convs and keeps their
//                   weights in registers as MFMA fragments for the whole kernel (3 x C/32 fragments)
//             A     stays in registers: the accumulator layout of S^T (a lane holds 4 consecutive pixels of one channel) IS the
//                   operand layout of the product over pixels
//             B     online softmax over pixels: running max / sum per channel, E = exp2(B - max) to LDS in operand layout,
//                   the rescale factor of the channel to LDS
//             V     computed in the transposed orientation (a lane holds 4 consecutive CHANNELS of one pixel); per-pixel softmax
//                   over the 128 channels = in-wave reduction + one (max, sum) pair per wave and pixel through LDS;
//                   leaves in 16 bit, token-major (the operand layout of pass 2) as whole 256-byte rows
//             G^T  += A E^T (wave w: rows 16w.. of G, all 128 columns, 32 accumulator registers)
//           the range's un-normalised G^T, max and sum go to the workspace
//   combine one small workgroup per image: merges the ranges' partial results (softmax merge), M' = WP G in 16 bit; with one range
//           per image (B >= number of CUs) pass 1 does this itself at its end
//           (y = WP (G V) + bP is evaluated as (WP G) V + bP: the (c_m x HW) intermediate of the reference's order disappears)
//   pass 2  y = M' V + bP: M' fragments in registers, V tiles by LDS-DMA (XOR-swizzled at the source), output staged through a
//           wave-private slab so that every store instruction writes whole 256-byte row segments of y
//
// HBM traffic per image: x once (fp32), V twice (16 bit, c_n / C of x each), y once -- 2.5x less than the unfused pipeline.
// Envelope: 16-bit operand modes, c_m = c_n = 128, C in {128, 256}, H*W a multiple of 4; everything else takes double_attn.hip.
#include "common.h"
#include "mma.h"

namespace {

constexpr int CM = 128;                       // c_m = c_n: 8 waves x 16 channels
constexpr int PT = 32;                        // pixels per tile of pass 1
constexpr int MAXS = 32;                      // pixel ranges per image (pass 1 workgroups per image)

__device__ __forceinline__ float ex2(float v) { return __builtin_amdgcn_exp2f(v); }

// Workgroup barrier of pass 1.  __syncthreads() carries a release fence that drains the vector-memory counter -- i.e. every LDS-DMA
// tile in flight -- at every barrier; here only the LDS queue is drained and the in-flight tiles are counted by hand (vmcnt(N)).
#define DA_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}

// One 1 KB LDS-DMA (16 bytes per lane, lane-linear at `dst`) issued BEHIND THE COMPILER'S BACK.  Through the builtin, hipcc knows an
// LDS-DMA is in flight and -- having no alias scopes for most LDS accesses -- puts `s_waitcnt vmcnt(0)` in front of some later,
// unrelated ds_read / ds_write (which one changed with every edit of pass 1): that wait drains the whole prefetch ring once per
// tile.  Pass 1 counts its own DMAs (vmcnt(N) before the barrier that publishes a tile), so the compiler does not need to know.
// m0 (the LDS-DMA destination base) belongs to the compiler: it is saved and restored INSIDE the statement instead of being named
// as a clobber (hipcc does not honour an "m0" clobber: "clobber list contains reserved registers"; cdna_hip_programming.md 5.7).
__device__ __forceinline__ void lds_dma16(const void* src, const void* dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr(dst)))
                 : "memory");
}

struct DaArgs {
    const float* x;            // (B, C, HW) fp32
    const void* w16;           // (3 * 128, C) 16 bit: WA | log2e WB | log2e WV
    const float* bias;         // (3 * 128) fp32, the B and V parts in log2 units
    void* v16;                 // (B, HW, 128) 16 bit: softmax over channels of V, token-major
    float* gt;                 // (B, S, 128 k', 128 m) fp32: un-normalised partial G^T
    float* ml;                 // (B, S, 2, 128): running max (log2 units) and sum of exp2 per channel of B
    const void* wp16;          // (C, 128) 16 bit
    void* m16;                 // (B, C, 128) 16 bit: M' = WP G
    const float* bp;           // (C)
    float* y;                  // (B, C, HW)
    int B, HW, S, tiles, tiles_per;
};

// ---- weights to the operand format (once per call; 130 K elements) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void da_prep_kernel(const float* __restrict__ wA, const float* __restrict__ wB, const float* __restrict__ wV,
                                                      const float* __restrict__ bA, const float* __restrict__ bB, const float* __restrict__ bV,
                                                      const float* __restrict__ wP, T* __restrict__ w16, float* __restrict__ bias,
                                                      T* __restrict__ wp16, int C) {
    constexpr float LOG2E = 1.4426950408889634f;
    const int i = blockIdx.x * 256 + threadIdx.x, n = CM * C;
    if (i < n) {
        w16[i] = (T)wA[i];
        w16[n + i] = (T)(wB[i] * LOG2E);
        w16[2 * n + i] = (T)(wV[i] * LOG2E);
        wp16[i] = (T)wP[i];                                         // (C, 128): the same element count
    }
    if (i < CM) {
        bias[i] = bA[i];
        bias[CM + i] = bB[i] * LOG2E;
        bias[2 * CM + i] = bV[i] * LOG2E;
    }
}

// ---- pass 1 ------------------------------------------------------------------------------------------------------------------------
template <int PREC, int KS>
__global__ __launch_bounds__(512) void da_pass1_kernel(const DaArgs a) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    static_assert(Mma<PREC>::NSPLIT == 1, "16-bit operand modes only");
    constexpr int C = 32 * KS;
    constexpr int XP = C + 8;                  // pitch of X^T rows (elements): 16 rows x 16 bytes land on 64 distinct banks
    constexpr int EP = 40;                     // pitch of E rows (32 px + 8)
    constexpr int VP = CM + 8;                 // pitch of the V staging rows
    constexpr int NDMA = C / 64;               // 1 KB LDS-DMA instructions per wave and tile (C / 8 channel blocks over 8 waves)
    constexpr int RB = 8 * PT + 32;            // floats per 8-channel block of the raw tile: 1 KB of data + 128 bytes of padding
    __shared__ __attribute__((aligned(16))) float s_raw[3][(C / 8) * RB];
    __shared__ __attribute__((aligned(16))) unsigned short s_xt[PT * XP];
    __shared__ __attribute__((aligned(16))) unsigned short s_e[CM * EP];
    __shared__ __attribute__((aligned(16))) unsigned short s_v[PT * VP];
    __shared__ float s_alpha[CM];
    __shared__ __attribute__((aligned(16))) float s_bias[3 * CM];
    __shared__ __attribute__((aligned(16))) unsigned s_flag[8];    // per wave: some channel of it moved its reference maximum this tile
    __shared__ __attribute__((aligned(8))) float s_stat[8][PT][2];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.S, sp = blockIdx.x - b * a.S;
    const int t0 = sp * a.tiles_per;
    const int n = min(a.tiles_per, a.tiles - t0);
    const float* xb = a.x + (long)b * C * a.HW;

    // wave w moves channel blocks w * NDMA + i: 8 channels x 32 px = 8 x 128-byte row pieces = 1 KB, lane-linear in LDS.  Pixels
    // past the image (last tile) re-read the row's last 16 bytes: finite duplicates of this image's data, masked below.
    // Lane l lands in 16-byte slot l of the block and fetches row l / 8, chunk (l % 8) ^ 4 (l / 32): with the 128-byte block padding
    // this makes the 4 x 4 transposing reads below conflict-free (a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, ...).
    auto dma = [&](int tile, int buf) {
        const int px = min(tile * PT + (((lane & 7) ^ ((lane >> 5) << 2)) * 4), a.HW - 4);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int blk = w * NDMA + i;
            const float* src = xb + (long)(blk * 8 + (lane >> 3)) * a.HW + px;
            lds_dma16(src, &s_raw[buf][blk * RB]);
        }
    };

    // this wave's slice of the three convs as MFMA fragments; either orientation of the product takes the same fragment
    v8 wf[3][KS];
    {
        const el* wg = static_cast<const el*>(a.w16);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][ks] = *reinterpret_cast<const v8*>(wg + (long)(j * CM + 16 * w + l15) * C + ks * 32 + g * 8);
    }
    if (t < 3 * CM) s_bias[t] = a.bias[t];                          // (visible after the barrier below)
    dma(t0, 0);                                                     // three tiles in flight: 96 KB per CU, 24 MB on the device
    if (n > 1) dma(t0 + 1, 1);
    if (n > 2) dma(t0 + 2, 2);

    f4 G[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) G[j] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, lsum = 0.f;

    auto store_v = [&](int tile) {                                  // s_v -> 256-byte token rows of v16, 16 bytes per thread
        const int px = tile * PT + (t >> 4);
        if (px < a.HW)
            *reinterpret_cast<uint4*>(static_cast<el*>(a.v16) + ((long)b * a.HW + px) * CM + (t & 15) * 8) =
                *reinterpret_cast<const uint4*>(&s_v[(t >> 4) * VP + (t & 15) * 8]);
    };

    if (n > 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");   // weights and tile 0 landed
    else if (n > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = -1;
    for (int it = 0; it < n; ++it) {
        const int tile = t0 + it, px0 = tile * PT;
        buf = buf == 2 ? 0 : buf + 1;
        // ---- raw (C x 32 px fp32) -> X^T (32 px x C, 16 bit): 4 channels x 4 pixels per thread ---------------------------------------
#pragma unroll
        for (int u0 = 0; u0 < 2 * C; u0 += 512) {
            const int u = u0 + t;
            if (2 * C >= 512 || u < 2 * C) {
                // lane -> (channel group cg: 8 per wave, fastest; pixel quad q): the 8-byte stores of 16 consecutive lanes then cover all
                // 32 banks once
                const int cg = (u >> 6) * 8 + (u & 7), q = (u >> 3) & 7;
                // (inline asm: a compiler-visible read of the LDS-DMA target would get a vmcnt(0) in front of it -- all tiles in flight)
                f4 r[4];
                const unsigned ra = lds_addr(&s_raw[buf][(cg >> 1) * RB + (8 * (4 * (cg & 1)) + (q ^ (4 * (cg & 1)))) * 4]);
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:128\n\tds_read_b128 %2, %4 offset:256\n\t"
                             "ds_read_b128 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(ra) : "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<v4*>(&s_xt[(4 * q + i) * XP + 4 * cg]) = M_::cvt(f4{r[0][i], r[1][i], r[2][i], r[3][i]});
            }
        }
        DA_BAR();                                                   // B1: X^T complete; raw[buf] and s_v of the previous tile free to reuse / read
        if (it > 0) store_v(tile - 1);
        if (it + 3 < n) dma(tile + 3, buf);

        // ---- S^T = X^T W^T (A, B: rows = pixels) and W X (V: rows = channels) ----------------------------------------------------------
        f4 acc[2][3];
        const float bA = s_bias[16 * w + l15], bB = s_bias[CM + 16 * w + l15];
        const f4 bV = *reinterpret_cast<const f4*>(&s_bias[2 * CM + 16 * w + 4 * g]);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {                            // accumulators start at the bias (column = channel for A / B, row for V)
            acc[rb][0] = f4{bA, bA, bA, bA};
            acc[rb][1] = f4{bB, bB, bB, bB};
            acc[rb][2] = bV;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const v8 xa = *reinterpret_cast<const v8*>(&s_xt[(16 * rb + l15) * XP + ks * 32 + g * 8]);
                acc[rb][0] = M_::mma(xa, wf[0][ks], acc[rb][0]);
                acc[rb][1] = M_::mma(xa, wf[1][ks], acc[rb][1]);
                acc[rb][2] = M_::mma(wf[2][ks], xa, acc[rb][2]);
            }
        const bool tail = px0 + PT > a.HW;                          // wave-uniform
        // A: operand of the product over pixels, k slot i <-> pixel 4g + i, slot 4 + i <-> pixel 16 + 4g + i
        v8 afrag;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            afrag[i] = M_::cvt1(acc[0][0][i]);
            afrag[4 + i] = M_::cvt1(acc[1][0][i]);
        }
        // B: online softmax over pixels of channel 16w + l15; V: lane holds channels 16w + 4g + [0,4) of pixels 16 rb + l15, softmax
        // over channels, first within the wave.  The three maxima (and later the two sums) cross the lane groups together: one LDS
        // round trip per step instead of one per value.
        f4 ev[2];
        {
            float e[8];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[4 * rb + i] = acc[rb][1][i];
                    if (tail && px0 + 16 * rb + 4 * g + i >= a.HW) e[4 * rb + i] = -INFINITY;
                }
            float mx = fmaxf(fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])), fmaxf(fmaxf(e[4], e[5]), fmaxf(e[6], e[7])));
            float mv0 = fmaxf(fmaxf(acc[0][2].x, acc[0][2].y), fmaxf(acc[0][2].z, acc[0][2].w));
            float mv1 = fmaxf(fmaxf(acc[1][2].x, acc[1][2].y), fmaxf(acc[1][2].z, acc[1][2].w));
            {
                const float a0 = __shfl_xor(mx, 16, WAVE), a1 = __shfl_xor(mv0, 16, WAVE), a2 = __shfl_xor(mv1, 16, WAVE);
                mx = fmaxf(mx, a0); mv0 = fmaxf(mv0, a1); mv1 = fmaxf(mv1, a2);
                const float b0 = __shfl_xor(mx, 32, WAVE), b1 = __shfl_xor(mv0, 32, WAVE), b2 = __shfl_xor(mv1, 32, WAVE);
                mx = fmaxf(mx, b0); mv0 = fmaxf(mv0, b1); mv1 = fmaxf(mv1, b2);
            }
            // the reference maximum of a channel moves only when the tile's maximum passes it by more than 2^8 (E <= 256 fits any 16-bit
            // format; G and the sum stay consistent because both are relative to the same reference): after the first tiles no
            // channel moves any more and the rescale of G below is skipped for the whole wave
            const bool move = mx > m_run + 8.0f;                    // true on the first tile (m_run = -inf)
            const float m_new = move ? mx : m_run;
            const float ms = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = move ? ex2(m_run - ms) : 1.0f;      // 0 on the first tile
            float es = 0.f;
            v8 ef;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x = ex2(e[k] - ms);
                es += x;
                ef[k] = M_::cvt1(x);
            }
            lsum = lsum * alpha + es;
            m_run = m_new;
            *reinterpret_cast<v8*>(&s_e[(16 * w + l15) * EP + 8 * g]) = ef;
            if (g == 0) s_alpha[16 * w + l15] = alpha;
            const bool any = __ballot(move) != 0;
            if (lane == 0) s_flag[w] = any ? 1u : 0u;
            ev[0] = f4{ex2(acc[0][2].x - mv0), ex2(acc[0][2].y - mv0), ex2(acc[0][2].z - mv0), ex2(acc[0][2].w - mv0)};
            ev[1] = f4{ex2(acc[1][2].x - mv1), ex2(acc[1][2].y - mv1), ex2(acc[1][2].z - mv1), ex2(acc[1][2].w - mv1)};
            float s0 = (ev[0].x + ev[0].y) + (ev[0].z + ev[0].w), s1 = (ev[1].x + ev[1].y) + (ev[1].z + ev[1].w);
            {
                const float a0 = __shfl_xor(s0, 16, WAVE), a1 = __shfl_xor(s1, 16, WAVE);
                s0 += a0; s1 += a1;
                const float b0 = __shfl_xor(s0, 32, WAVE), b1 = __shfl_xor(s1, 32, WAVE);
                s0 += b0; s1 += b1;
            }
            if (g == 0) {
                s_stat[w][l15][0] = mv0;
                s_stat[w][l15][1] = s0;
                s_stat[w][16 + l15][0] = mv1;
                s_stat[w][16 + l15][1] = s1;
            }
        }
        if (it + 3 < n)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");   // tile it+1 landed; it+2, it+3 may still fly
        else if (it + 2 < n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DA_BAR();                                                   // B2: E, alpha, V statistics of every wave visible; raw of tile it+1 complete

        // ---- G^T rows 16w.. += A E^T -----------------------------------------------------------------------------------------------
        {
            const uint4 f0 = *reinterpret_cast<const uint4*>(&s_flag[0]), f1 = *reinterpret_cast<const uint4*>(&s_flag[4]);
            if (__builtin_amdgcn_readfirstlane((f0.x | f0.y | f0.z | f0.w) | (f1.x | f1.y | f1.z | f1.w))) {
#pragma unroll
                for (int j = 0; j < 8; ++j) G[j] = G[j] * s_alpha[16 * j + l15];
            }
        }
        // (inline asm for the same reason as the raw-tile reads: the compiler puts a vmcnt(0) in front of a visible read of s_e)
        {
            const unsigned ea = lds_addr(&s_e[l15 * EP + 8 * g]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v8 ef[4];
                if (h == 0)
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1280\n\tds_read_b128 %2, %4 offset:2560\n\t"
                                 "ds_read_b128 %3, %4 offset:3840\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(ef[0]), "=&v"(ef[1]), "=&v"(ef[2]), "=&v"(ef[3]) : "v"(ea) : "memory");
                else
                    asm volatile("ds_read_b128 %0, %4 offset:5120\n\tds_read_b128 %1, %4 offset:6400\n\tds_read_b128 %2, %4 offset:7680\n\t"
                                 "ds_read_b128 %3, %4 offset:8960\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(ef[0]), "=&v"(ef[1]), "=&v"(ef[2]), "=&v"(ef[3]) : "v"(ea) : "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) G[4 * h + j] = M_::mma(afrag, ef[j], G[4 * h + j]);
            }
        }
        static_assert(16 * EP * 2 == 1280, "immediate offsets of the E reads");
        // ---- V: merge the eight waves' (max, sum) per pixel, scale, park in s_v (token-major) -----------------------------------------
        // lane -> (pixel = lane & 31, half of the waves = lane >> 5); the factor of this wave's channels at each pixel is then handed
        // to the lanes that hold that pixel's values
        float fpx;
        {
            const int px = lane & 31, h = lane >> 5;
            float sm[4], ss[4], M = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 st = *reinterpret_cast<const float2*>(&s_stat[4 * h + q][px][0]);
                sm[q] = st.x;
                ss[q] = st.y;
                M = fmaxf(M, st.x);
            }
            float den = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) den += ss[q] * ex2(sm[q] - M);
            const float Mo = __shfl_xor(M, 32, WAVE), deno = __shfl_xor(den, 32, WAVE);
            const float Mt = fmaxf(M, Mo);
            den = den * ex2(M - Mt) + deno * ex2(Mo - Mt);
            fpx = ex2(s_stat[w][px][0] - Mt) * __builtin_amdgcn_rcpf(den);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float f = __shfl(fpx, 16 * rb + l15, WAVE);
            *reinterpret_cast<v4*>(&s_v[(16 * rb + l15) * VP + 16 * w + 4 * g]) = M_::cvt(ev[rb] * f);
        }
    }
    __syncthreads();
    store_v(t0 + n - 1);
    lsum += __shfl_xor(lsum, 16, WAVE);
    lsum += __shfl_xor(lsum, 32, WAVE);
    if (a.S == 1) {
        // ---- the whole image was this workgroup's: normalise G and form M' = WP G here (what da_combine_kernel does for S > 1) -------
        constexpr int GP = CM + 8;
        static_assert(CM * GP * 2 <= (int)sizeof(s_raw), "G^T (16 bit) is parked in the raw tile ring");
        unsigned short* s_g = reinterpret_cast<unsigned short*>(&s_raw[0][0]);
        if (g == 0) s_alpha[16 * w + l15] = 1.0f / lsum;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<v4*>(&s_g[(16 * j + l15) * GP + 16 * w + 4 * g]) = M_::cvt(G[j] * s_alpha[16 * j + l15]);
        __syncthreads();
        const el* wp = static_cast<const el*>(a.wp16);
        el* m16 = static_cast<el*>(a.m16) + (long)b * C * CM;
        for (int ct = w; ct < C / 16; ct += 8) {
            v8 bf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const v8*>(wp + (long)(16 * ct + l15) * CM + 32 * ks + 8 * g);
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = M_::mma(*reinterpret_cast<const v8*>(&s_g[(16 * rt + l15) * GP + 32 * ks + 8 * g]), bf[ks], acc);
                *reinterpret_cast<v4*>(m16 + (long)(16 * ct + l15) * CM + 16 * rt + 4 * g) = M_::cvt(acc);
            }
        }
        return;
    }
    // ---- the range's partial result ------------------------------------------------------------------------------------------------
    float* gt = a.gt + (long)(b * a.S + sp) * CM * CM;
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f4*>(gt + (long)(16 * j + l15) * CM + 16 * w + 4 * g) = G[j];
    if (g == 0) {
        float* ml = a.ml + (long)(b * a.S + sp) * 2 * CM;
        ml[16 * w + l15] = m_run;
        ml[CM + 16 * w + l15] = lsum;
    }
}

// ---- combine: G = merge of the ranges (normalised), M'^T = G^T WP^T ----------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void da_combine_kernel(const DaArgs a, int C) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using v4 = typename M_::v4;
    using el = typename M_::e;
    constexpr int GP = CM + 8;
    __shared__ float s_sc[MAXS][CM];
    __shared__ __attribute__((aligned(16))) unsigned short s_g[CM * GP];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4, b = blockIdx.x, S = a.S;
    if (t < CM) {
        const float* ml = a.ml + (long)b * S * 2 * CM;
        float M = -INFINITY;
        for (int s = 0; s < S; ++s) M = fmaxf(M, ml[s * 2 * CM + t]);
        float L = 0.f;
        for (int s = 0; s < S; ++s) L += ml[s * 2 * CM + CM + t] * ex2(ml[s * 2 * CM + t] - M);
        const float inv = 1.0f / L;
        for (int s = 0; s < S; ++s) s_sc[s][t] = ex2(ml[s * 2 * CM + t] - M) * inv;
    }
    __syncthreads();
    const float* gt = a.gt + (long)b * S * CM * CM;
#pragma unroll 4
    for (int e = 0; e < 16; ++e) {
        const int idx = t + 256 * e, k = idx >> 5, mq = idx & 31;
        f4 sum = f4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) sum += *reinterpret_cast<const f4*>(gt + ((long)s * CM + k) * CM + 4 * mq) * s_sc[s][k];
        *reinterpret_cast<v4*>(&s_g[k * GP + 4 * mq]) = M_::cvt(sum);
    }
    __syncthreads();
    const el* wp = static_cast<const el*>(a.wp16);
    el* m16 = static_cast<el*>(a.m16) + (long)b * C * CM;
    for (int ct = w; ct < C / 16; ct += 4) {
        v8 bf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const v8*>(wp + (long)(16 * ct + l15) * CM + 32 * ks + 8 * g);
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = M_::mma(*reinterpret_cast<const v8*>(&s_g[(16 * rt + l15) * GP + 32 * ks + 8 * g]), bf[ks], acc);
            // lane: column o = 16 ct + l15, rows k' = 16 rt + 4g + [0,4)
            *reinterpret_cast<v4*>(m16 + (long)(16 * ct + l15) * CM + 16 * rt + 4 * g) = M_::cvt(acc);
        }
    }
}

// ---- pass 2: y = M' V + bP -----------------------------------------------------------------------------------------------------------
template <int PREC, int CT>                    // CT = C / 128: output-channel tiles per wave
__global__ __launch_bounds__(512, 2) void da_pass2_kernel(const DaArgs a, int nsub, int groups) {
    using M_ = Mma<PREC>;
    using v8 = typename M_::v8;
    using el = typename M_::e;
    constexpr int C = 128 * CT, PX = 64, SP = PX + 4;               // slab pitch (floats): 16 rows x 16 bytes on 64 distinct banks
    __shared__ __attribute__((aligned(16))) unsigned short s_vt[2][PX * CM];
    __shared__ __attribute__((aligned(16))) float s_slab[8][16 * SP];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
    const int id = xcd_contiguous_block();
    const int b = id / groups, grp = id - b * groups;
    const el* vg = static_cast<const el*>(a.v16) + (long)b * a.HW * CM;
    // V tile: 64 token rows x 256 bytes; DMA instruction i covers rows 4i..4i+3, lane-linear in LDS.  The 16-byte chunk at LDS
    // position c of row r holds source chunk c ^ (r & 15): the operand reads of 16 consecutive rows then hit 16 distinct slots.
    auto dma = [&](int sub, int buf) {
        const int px0 = (grp * nsub + sub) * PX;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 2 * w + i, r = 4 * blk + (lane >> 4);
            const int px = min(px0 + r, a.HW - 1);
            const el* src = vg + (long)px * CM + ((lane & 15) ^ (r & 15)) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&s_vt[buf][blk * 4 * CM], 16, 0, 0);
        }
    };
    for (int sub = 0; sub < nsub && sub < 2; ++sub)
        if ((grp * nsub + sub) * PX < a.HW) dma(sub, sub);
    v8 mf[CT][4];
    float bp[CT];
    {
        const el* mg = static_cast<const el*>(a.m16) + (long)b * C * CM;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int o = 16 * (CT * w + ct) + l15;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mf[ct][ks] = *reinterpret_cast<const v8*>(mg + (long)o * CM + 32 * ks + 8 * g);
            bp[ct] = a.bp[o];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* slab = s_slab[w];
    for (int sub = 0; sub < nsub; ++sub) {
        const int px0 = (grp * nsub + sub) * PX;
        if (px0 >= a.HW) break;
        const int buf = sub & 1;
        if (sub >= 2) {                                             // (nsub <= 2 in every launch below; kept general)
            __syncthreads();
            dma(sub, buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        f4 acc[4][CT];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const v8 va = *reinterpret_cast<const v8*>(&s_vt[buf][(16 * rb + l15) * CM + (((4 * ks + g) ^ l15) * 8)]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[rb][ct] = M_::mma(va, mf[ct][ks], acc[rb][ct]);
            }
        // lane: column o = l15 of tile ct, rows px = 16 rb + 4g + [0,4) -> slab [o][px] -> 256-byte row segments of y
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) *reinterpret_cast<f4*>(&slab[l15 * SP + 16 * rb + 4 * g]) = acc[rb][ct] + bp[ct];
            __builtin_amdgcn_wave_barrier();                       // lanes exchange through the wave's own slab: keep the compiler from
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // moving a lane's reads above its (different-address) writes
            float* yo = a.y + ((long)b * C + 16 * (CT * w + ct)) * a.HW + px0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * i + (lane >> 4), c4 = (lane & 15) * 4;
                const f4 v = *reinterpret_cast<const f4*>(&slab[row * SP + c4]);
                if (px0 + c4 < a.HW) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(yo + (long)row * a.HW + c4));
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

inline size_t r256(size_t n) { return (n + 255) & ~(size_t)255; }

int plan_ranges(int B, int tiles) {
    const int ncu = mi355::resident_slots(1);
    int S = (ncu + B - 1) / B;
    if (mi355::opt_da_ranges() > 0) S = (int)mi355::opt_da_ranges();   // pinned: the summation order over pixels no longer depends on B
    if (S > MAXS) S = MAXS;
    if (S > tiles / 4 && mi355::opt_da_ranges() == 0) S = tiles / 4;
    if (S > tiles) S = tiles;
    if (S < 1) S = 1;
    const int per = (tiles + S - 1) / S;
    return (tiles + per - 1) / per;                                  // no empty range
}

}  // namespace

namespace mi355 {

bool double_attn_fused_ok(int B, int C, int cm, int cn, int HW, int precision) {
    return (precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16) && cm == CM && cn == CM && (C == 128 || C == 256) &&
           (HW & 3) == 0 && HW >= 4 && (long)B * MAXS < (1l << 24) && (long)HW * CM < (1l << 30);
}

size_t double_attn_fused_workspace(int B, int C, int HW) {
    const int tiles = (HW + PT - 1) / PT, S = plan_ranges(B, tiles);
    return r256((size_t)3 * CM * C * 2) + r256((size_t)3 * CM * 4) + r256((size_t)C * CM * 2) + r256((size_t)B * HW * CM * 2) +
           r256((size_t)B * S * CM * CM * 4) + r256((size_t)B * S * 2 * CM * 4) + r256((size_t)B * C * CM * 2);
}

int double_attn_fused(const float* x, const float* wA, const float* bA, const float* wB, const float* bB, const float* wV, const float* bV,
                      const float* wP, const float* bP, float* y, int B, int C, int HW, int precision, void* ws, hipStream_t st) {
    DaArgs a;
    a.tiles = (HW + PT - 1) / PT;
    a.S = plan_ranges(B, a.tiles);
    a.tiles_per = (a.tiles + a.S - 1) / a.S;
    a.B = B;
    a.HW = HW;
    char* p = static_cast<char*>(ws);
    void* w16 = p;                             p += r256((size_t)3 * CM * C * 2);
    float* bias = reinterpret_cast<float*>(p); p += r256((size_t)3 * CM * 4);
    void* wp16 = p;                            p += r256((size_t)C * CM * 2);
    a.v16 = p;                                 p += r256((size_t)B * HW * CM * 2);
    a.gt = reinterpret_cast<float*>(p);        p += r256((size_t)B * a.S * CM * CM * 4);
    a.ml = reinterpret_cast<float*>(p);        p += r256((size_t)B * a.S * 2 * CM * 4);
    a.m16 = p;
    a.x = x; a.w16 = w16; a.bias = bias; a.wp16 = wp16; a.bp = bP; a.y = y;
    const int pgrid = cdiv((long)CM * C, 256);
    const int nsub = 2, groups = cdiv(HW, 64 * nsub);
    if (precision == MI355_PREC_FP16) {
        da_prep_kernel<_Float16><<<pgrid, 256, 0, st>>>(wA, wB, wV, bA, bB, bV, wP, static_cast<_Float16*>(w16), bias, static_cast<_Float16*>(wp16), C);
        if (C == 256) da_pass1_kernel<1, 8><<<B * a.S, 512, 0, st>>>(a);
        else          da_pass1_kernel<1, 4><<<B * a.S, 512, 0, st>>>(a);
        if (a.S > 1) da_combine_kernel<1><<<B, 256, 0, st>>>(a, C);
        if (C == 256) da_pass2_kernel<1, 2><<<B * groups, 512, 0, st>>>(a, nsub, groups);
        else          da_pass2_kernel<1, 1><<<B * groups, 512, 0, st>>>(a, nsub, groups);
    } else {
        da_prep_kernel<__bf16><<<pgrid, 256, 0, st>>>(wA, wB, wV, bA, bB, bV, wP, static_cast<__bf16*>(w16), bias, static_cast<__bf16*>(wp16), C);
        if (C == 256) da_pass1_kernel<2, 8><<<B * a.S, 512, 0, st>>>(a);
        else          da_pass1_kernel<2, 4><<<B * a.S, 512, 0, st>>>(a);
        if (a.S > 1) da_combine_kernel<2><<<B, 256, 0, st>>>(a, C);
        if (C == 256) da_pass2_kernel<2, 2><<<B * groups, 512, 0, st>>>(a, nsub, groups);
        else          da_pass2_kernel<2, 1><<<B * groups, 512, 0, st>>>(a, nsub, groups);
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // namespace mi355
