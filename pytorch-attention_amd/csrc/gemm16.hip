// gemm16.hip -- the fast GEMM path: 16-bit operands resident in HBM, LDS-DMA staging, MFMA 16x16x32, fused epilogue.
//
//   Y (M x N) = resid + gamma * act( X16 (M x K) . W16^T (N x K) + bias )        X16, W16: IEEE half (precision 1) or bfloat16 (2)
//
// Why a second GEMM: with fp32 operands in HBM the 128x128 tile of gemm.hip moves 32 KB per 1 MFLOP through L2 (32 FLOP/B) and
// tops out near 370 TFLOP/s -- L2-bandwidth bound (profiles/r01_*).  Keeping activations and weights in their MFMA operand
// format in memory halves those bytes, removes the convert + ds_write pass, and lets the tile go straight from HBM/L2 into LDS
// with `global_load_lds` (16 B per lane, no VGPR round trip).  The rounding point is unchanged (operands are rounded to 16 bit
// exactly once, accumulate and epilogue stay fp32), so results are bit-identical to gemm.hip in the same precision mode.
//
// Tile BM x BN x 64 (default 128 x 256, 8 waves of 64 x 64; variants in the launcher).  LDS rows are 128 B and UNPADDED because
// the DMA writes lane-linear; bank conflicts of the fragment reads are removed by an XOR swizzle applied on the SOURCE address
// (chunk ^= row & 7) and again on the ds_read (cdna_hip_programming.md rule 21).  One LDS buffer + two barriers per K-step and
// three resident workgroups per CU (24 waves) measured faster than double buffering at two workgroups per CU.
#include <type_traits>
#include "gemm16.h"

namespace {
using namespace g16;

// Tile BM x BN x 64 per workgroup of WM x WN waves; each wave owns a (BM/WM) x (BN/WN) sub-tile = MF x NF MFMA 16x16 tiles.
// TR = true: the product is written TRANSPOSED per image -- row m = img * tr_rows + c, column n -> Y[(img * N + n) * tr_rows + c]
// (+ the residual at the same address, bias indexed by n): the Mixer token-mixing product computed channel-major lands back in
// the token-major activation without a transpose pass.
template <typename T, bool OUT16, int BM, int BN, int WM, int WN, bool PRIO, int STAGES, bool TR = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm16_kernel(const G16Args g) {
    using v8 = typename Vec8<T>::t;
    using v4 = typename Vec8<T>::t4;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MF = TM / 16, NF = TN / 16;
    constexpr int STAGE = (BM + BN) * BK;                       // elements per stage
    constexpr int IA = BM / 8 / NW, IB = BN / 8 / NW;           // LDS-DMA instructions per wave per stage (8 rows each)
    constexpr int LDS_BYTES = STAGES * STAGE * 2;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && TM % 32 == 0 && TN % 16 == 0, "tile/wave geometry");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int tiles_n = (g.N + BN - 1) / BN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);

    // ---- LDS-DMA source pointers: each instruction fills 8 rows of 128 B; lane -> (row = lane >> 3, physical chunk = lane & 7)
    //      holds logical chunk (lane & 7) ^ (row & 7) ----------------------------------------------------------------------------
    const int lrow = lane >> 3, pch = lane & 7;
    const int csw = (pch ^ lrow) * 8;                           // row & 7 == lrow: every 8-row group starts at a multiple of 8
    const T* a_src[IA];
    const T* b_src[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        int ma = m0 + (wave * IA + i) * 8 + lrow; if (ma >= g.M) ma = g.M - 1;   // clamp: tail rows are computed and discarded
        a_src[i] = A + (long)ma * g.lda + csw;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        int nb = n0 + (wave * IB + i) * 8 + lrow; if (nb >= g.N) nb = g.N - 1;
        b_src[i] = B + (long)nb * g.ldb + csw;
    }
    auto issue = [&](int stage, int k0) {
        T* sA = lds + stage * STAGE + (wave * IA * 8) * BK;
        T* sB = lds + stage * STAGE + BM * BK + (wave * IB * 8) * BK;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sA + i * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sB + i * 8 * BK), 16, 0, 0);
    };

    f4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
    auto compute = [&](int st) {
        const T* sA = lds + st * STAGE + (wr * TM + frow) * BK;
        const T* sB = lds + st * STAGE + BM * BK + (wc * TN + frow) * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = (((ks * 4 + fq) ^ fsw) * 8);
            v8 fa[MF], fb[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) fb[j] = *reinterpret_cast<const v8*>(sB + j * 16 * BK + off);
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[i] = *reinterpret_cast<const v8*>(sA + i * 16 * BK + off);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if constexpr (TR) acc[i][j] = mma16<T>(fa[i], fb[j], acc[i][j]);          // D = X.W^T: lane holds 4 rows of one column
                    else              acc[i][j] = mma16<T>(fb[j], fa[i], acc[i][j]);          // D^T = W.X^T: see the epilogue
                }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };
    if constexpr (STAGES == 2) {          // double-buffered: next tile's DMA in flight under this tile's MFMAs, one barrier per K-step
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int st = kt & 1;
            if (kt + 1 < nk) issue(st ^ 1, (kt + 1) * BK);
            compute(st);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if constexpr (STAGES >= 3) {
        // Ring of STAGES K-tiles for a LONE small workgroup per CU (the left-over rows of the two-accumulator kernel, see
        // linear16_dispatch): STAGES - 1 K-tiles of LDS-DMA stay in flight, counted s_waitcnt, ONE raw barrier per K-step.  The
        // barrier of step kt says every wave's share of K-tile kt has landed AND every wave has issued the MFMAs of step kt - 1 (its
        // fragment reads of that slot have returned), so the slot of K-tile kt - 1 is refilled right behind it.
        constexpr int AHEAD = STAGES - 1, STEADY = (IA + IB) * (STAGES - 2);
        static_assert(STEADY <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (s < nk) issue(s, s * BK);
        int slot = 0, fill = AHEAD;                             // slot = kt % STAGES, fill = (kt + AHEAD) % STAGES
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY) : "memory");
            else                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + AHEAD < nk) issue(fill, (kt + AHEAD) * BK);
            compute(slot);
            slot = slot + 1 == STAGES ? 0 : slot + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
    } else {                               // single buffer, two barriers per K-step: half the LDS, twice the resident workgroups --
        for (int kt = 0; kt < nk; ++kt) {  // load/compute overlap comes from the other workgroups on the CU
            issue(0, kt * BK);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    // ---- epilogue straight from the accumulators.  The MFMAs above compute the TRANSPOSED 16x16 tiles (weights as the A
    //      operand, activations as B), so lane (l15, g) holds output row m = i*16 + l15 and four CONSECUTIVE columns
    //      n = j*16 + g*4 + [0,4): one 16-byte (fp32) / 8-byte (16-bit) store per tile and lane, the four lane groups of a row
    //      completing a 64-byte run -- no LDS transpose, ~8x fewer epilogue instructions than the slab version -----------------
    if constexpr (TR) {
        // lane (l15, g) holds column n = j*16 + l15 and four CONSECUTIVE rows m = i*16 + g*4 + [0,4) -- consecutive channels of
        // one image (tr_rows % 4 == 0), i.e. 16 contiguous bytes of the token-major output; the four lane groups and the MF
        // row tiles of a wave complete a 256-byte run per token.  (Routing the tile through a per-wave LDS slab so that every
        // store instruction covers whole 256-byte runs was measured SLOWER: 494 -> 647 us on the K = 256 DoubleAttention product.)
        float* Cf = static_cast<float*>(g.C);
        const int l15 = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = n0 + wc * TN + j * 16 + l15;
            if (n >= g.N) continue;
            const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m0 + wr * TM + i * 16 + g4;
                if (m >= g.M) continue;                         // M % 4 == 0 is a launch precondition
                const int img = m / g.tr_rows, c = m - img * g.tr_rows;
                const long o = ((long)img * g.N + n) * g.tr_rows + c;
                f4 v = acc[i][j] + f4{bn, bn, bn, bn};
                if (g.act == MI355_ACT_GELU) v = gelu_fast4(v);
                if (g.resid) v = v + *reinterpret_cast<const f4*>(g.resid + o);
                *reinterpret_cast<f4*>(Cf + o) = v;
            }
        }
    } else {
        float* Cf = static_cast<float*>(g.C);
        T* Ch = static_cast<T*>(g.C);
        const int l15 = lane & 15, g4 = (lane >> 4) * 4;
        float rgmax = 0.f;                                     // fp16 range guard (common.h)
        f4 bias4[NF], gam4[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = n0 + wc * TN + j * 16 + g4;
            bias4[j] = (g.bias && n < g.N) ? *reinterpret_cast<const f4*>(g.bias + n) : f4{0.f, 0.f, 0.f, 0.f};
            gam4[j] = (g.gamma && n < g.N) ? *reinterpret_cast<const f4*>(g.gamma + n) : f4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + wr * TM + i * 16 + l15;
            if (m >= g.M) continue;
            const int mr = g.resid_period ? m % g.resid_period : m;      // periodic residual table (patch embedding)
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int n = n0 + wc * TN + j * 16 + g4;
                if (n >= g.N) continue;                         // N % 4 == 0 is a launch precondition
                f4 v = acc[i][j] + bias4[j];
                if (g.act == MI355_ACT_GELU) v = gelu_out4<OUT16>(v);
                if (g.gamma) v = v * gam4[j];
                if (g.resid) v = v + *reinterpret_cast<const f4*>(g.resid + (long)mr * g.ldc + n);
                if constexpr (OUT16) {
                    if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4(rgmax, v);
                    *reinterpret_cast<v4*>(Ch + (long)m * g.ldc + n) = v4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                } else {
                    *reinterpret_cast<f4*>(Cf + (long)m * g.ldc + n) = v;
                }
            }
        }
        if constexpr (OUT16 && std::is_same<T, _Float16>::value) rg_report(rgmax, g.ovf, 3u);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight-stationary streaming GEMM for SHORT K (64 / 128: CSWin stages 1-2, where every Linear is HBM-bound): a workgroup parks
// its BN x K slice of W in LDS once and then walks 128-row tiles of X -- A tiles double-buffered by LDS-DMA, one barrier pair per
// tile, direct-store epilogue -- so X is read once, Y written once and W is never re-fetched per tile (the generic kernel re-loads
// W for every 128 rows and exposes a full load->barrier->compute->store latency chain per tile: 2.2-2.6x off the streaming bound).
// 8 waves; wave grid (8/WN) x WN with WN = BN/64, i.e. every wave owns (128*WN/8) rows x 64 columns.
// LDS images are [k-chunk of 64][row][64] panels with the same source-side XOR swizzle as above.
// ---------------------------------------------------------------------------------------------------------------------------
// LNA = true: the row operand is the LayerNorm of fp32 rows (K = the normalised width): a thread loads a quarter of a row, the four
// lanes of a row reduce mean and variance, and the normalised 16-bit values go into the same swizzled LDS image the DMA would have
// produced.  The LayerNorm affine part is expected folded into W / bias by the caller (W' = W diag(gamma), b' = b + W beta).
template <typename T, bool OUT16, int BN, int KK, bool LNA = false>
__global__ __launch_bounds__(512, (LNA ? 2 : (KK == 64 ? 4 : 2))) void gemm16_ws_kernel(const G16Args g, int workers) {
    using v8 = typename Vec8<T>::t;
    using v4 = typename Vec8<T>::t4;
    constexpr int BM = 128, KC = KK / 64;
    constexpr int WN = BN / 64, WM = 8 / WN, TM = BM / WM, MF = TM / 16, NF = 4;
    constexpr int W_EL = BN * KK, A_EL = BM * KK;
    constexpr int SP = 64;                                      // 16-bit output slab: 16 rows x 64 columns per wave, 16-byte chunks XOR-swizzled by row
    constexpr int SLAB_EL = OUT16 ? 8 * 16 * SP : 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[(W_EL + 2 * A_EL + SLAB_EL) * 2];
    T* sW = reinterpret_cast<T*>(lds_raw);
    T* sAb = sW + W_EL;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    T* slab = reinterpret_cast<T*>(lds_raw) + W_EL + 2 * A_EL + wave * 16 * SP;
    const int nt = blockIdx.x % tiles_n, worker = blockIdx.x / tiles_n;
    const int n0 = nt * BN;
    const T* __restrict__ A = static_cast<const T*>(g.A);
    const T* __restrict__ B = static_cast<const T*>(g.B);
    const int lrow = lane >> 3, pch = lane & 7, csw = (pch ^ lrow) * 8;

    // W slice: BN*KC groups of 8 rows x 128 B, round-robin over the 8 waves
    for (int i = wave; i < (BN / 8) * KC; i += 8) {
        const int kc = i / (BN / 8), rg = i % (BN / 8);
        int nb = n0 + rg * 8 + lrow; if (nb >= g.N) nb = g.N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (long)nb * g.ldb + kc * 64 + csw),
                                         (__attribute__((address_space(3))) void*)(sW + (kc * BN + rg * 8) * 64), 16, 0, 0);
    }
    auto issue_a = [&](int buf, int mt) {
        const int m0 = mt * BM;
        for (int i = wave; i < (BM / 8) * KC; i += 8) {
            const int kc = i / (BM / 8), rg = i % (BM / 8);
            int ma = m0 + rg * 8 + lrow; if (ma >= g.M) ma = g.M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)ma * g.lda + kc * 64 + csw),
                                             (__attribute__((address_space(3))) void*)(sAb + buf * A_EL + (kc * BM + rg * 8) * 64), 16, 0, 0);
        }
    };

    // LNA staging: thread -> (row = t / 4, quarter of the row); PT floats per thread
    constexpr int PT = BM * KK / 512;
    const int arow = t >> 2, aq = t & 3;
    f4 xr[LNA ? PT / 4 : 1];
    auto load_a = [&](int mt) {
        int ma = mt * BM + arow; if (ma >= g.M) ma = g.M - 1;
        const float* p = g.Af + (long)ma * g.lda + aq * PT;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) xr[i] = *reinterpret_cast<const f4*>(p + i * 4);
    };
    auto commit_a = [&](int buf) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) s += (xr[i].x + xr[i].y) + (xr[i].z + xr[i].w);
        s += __shfl_xor(s, 1, WAVE);
        s += __shfl_xor(s, 2, WAVE);
        const float mean = s * (1.0f / (float)KK);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PT / 4; ++i) {
            const f4 d = xr[i] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        q += __shfl_xor(q, 1, WAVE);
        q += __shfl_xor(q, 2, WAVE);
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)KK) + g.ln_eps);
#pragma unroll
        for (int j = 0; j < PT / 8; ++j) {
            const int k0 = aq * PT + j * 8, kc = k0 / 64, ch = (k0 % 64) / 8;
            const f4 a0 = (xr[2 * j] - mean) * rstd, a1 = (xr[2 * j + 1] - mean) * rstd;
            *reinterpret_cast<v8*>(sAb + buf * A_EL + (kc * BM + arow) * 64 + ((ch ^ (arow & 7)) * 8)) =
                v8{(T)a0.x, (T)a0.y, (T)a0.z, (T)a0.w, (T)a1.x, (T)a1.y, (T)a1.z, (T)a1.w};
        }
    };

    const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7, l15 = frow, g4 = fq * 4;
    float* Cf = static_cast<float*>(g.C);
    T* Ch = static_cast<T*>(g.C);
    float rgmax = 0.f;                                         // fp16 range guard over every 16-bit value this thread converts

    int mt = worker;
    if constexpr (LNA) {
        if (mt < tiles_m) { load_a(mt); commit_a(0); }
    } else {
        if (mt < tiles_m) issue_a(0, mt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; mt < tiles_m; mt += workers, ++it) {
        const int cur = it & 1;
        if constexpr (LNA) {
            if (mt + workers < tiles_m) load_a(mt + workers);                    // next tile's rows fly into registers under this tile's math
        } else {
            if (mt + workers < tiles_m) issue_a(cur ^ 1, mt + workers);          // next tile flies under this tile's math
        }
        f4 acc[MF][NF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const T* pa = sAb + cur * A_EL + (kc * BM + wr * TM + frow) * 64;
            const T* pb = sW + (kc * BN + wc * 64 + frow) * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = (((ks * 4 + fq) ^ fsw) * 8);
                v8 fa[MF], fb[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) fb[j] = *reinterpret_cast<const v8*>(pb + j * 16 * 64 + off);
#pragma unroll
                for (int i = 0; i < MF; ++i) fa[i] = *reinterpret_cast<const v8*>(pa + i * 16 * 64 + off);
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = mma16<T>(fb[j], fa[i], acc[i][j]);   // transposed tiles, see above
            }
        }
        if constexpr (LNA) {
            if (mt + workers < tiles_m) commit_a(cur ^ 1);                        // normalise + park the next tile (buffer cur^1 was last read one barrier ago)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // next A tile landed (and older stores retired)
        __syncthreads();                                                      // ... for every wave; buffer `cur` is free again
        const int m0 = mt * BM;
        if constexpr (OUT16) {
            // 16-bit output: 8-byte per-lane stores would touch each 128-byte line four times; go through a per-wave LDS slab
            // (16 rows x 64 columns) instead and write whole 128-byte rows, 16 bytes per lane
#pragma unroll
            for (int i = 0; i < MF; ++i) {
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int n = n0 + wc * 64 + j * 16 + g4;
                    f4 v = acc[i][j];
                    if (n < g.N) {
                        if (g.bias) v = v + *reinterpret_cast<const f4*>(g.bias + n);
                        if (g.act == MI355_ACT_GELU) v = gelu_out4<OUT16>(v);
                        if (g.gamma) v = v * *reinterpret_cast<const f4*>(g.gamma + n);
                        const int m = m0 + wr * TM + i * 16 + l15;
                        if (g.resid && m < g.M) v = v + *reinterpret_cast<const f4*>(g.resid + (long)m * g.ldc + n);
                    }
                    if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4(rgmax, v);
                    *reinterpret_cast<v4*>(slab + l15 * SP + (((j * 2 + (fq >> 1)) ^ (l15 & 7)) * 8) + (fq & 1) * 4) =
                        v4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rl = h * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                    const int m = m0 + wr * TM + i * 16 + rl, n = n0 + wc * 64 + c8;
                    if (m < g.M && n < g.N)                           // N % 8 == 0 on this path (launcher)
                        *reinterpret_cast<v8*>(Ch + (long)m * g.ldc + n) =
                            *reinterpret_cast<const v8*>(slab + rl * SP + (((lane & 7) ^ (rl & 7)) * 8));
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        } else {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + wr * TM + i * 16 + l15;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int n = n0 + wc * 64 + j * 16 + g4;
                if (n >= g.N) continue;
                f4 v = acc[i][j];
                if (g.bias) v = v + *reinterpret_cast<const f4*>(g.bias + n);       // L1-resident, reloaded per tile to keep VGPRs <= 128
                if (g.act == MI355_ACT_GELU) v = gelu_fast4(v);
                if (g.gamma) v = v * *reinterpret_cast<const f4*>(g.gamma + n);
                if (g.resid) v = v + *reinterpret_cast<const f4*>(g.resid + (long)m * g.ldc + n);
                *reinterpret_cast<f4*>(Cf + (long)m * g.ldc + n) = v;
            }
        }
        }
    }
    if constexpr (OUT16 && std::is_same<T, _Float16>::value) rg_report(rgmax, g.ovf, 3u);
}

// fp32 -> 16-bit operand format, 8 elements per thread (2 x 16-B loads, 1 x 16-B store)
template <typename T>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ src, T* __restrict__ dst, long n8, long n, unsigned* ovf) {
    using v8 = typename Vec8<T>::t;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    float rgmax = 0.f;                                         // fp16 range guard (common.h): free in a bandwidth-bound pass
    for (; i < n8; i += stride) {
        const f4 a = reinterpret_cast<const f4*>(src)[2 * i], b = reinterpret_cast<const f4*>(src)[2 * i + 1];
        if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax4(rg_absmax4(rgmax, a), b);
        reinterpret_cast<v8*>(dst)[i] = v8{(T)a.x, (T)a.y, (T)a.z, (T)a.w, (T)b.x, (T)b.y, (T)b.z, (T)b.w};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const float v = src[n8 * 8 + threadIdx.x];
        if constexpr (std::is_same<T, _Float16>::value) rgmax = rg_absmax1(rgmax, v);
        dst[n8 * 8 + threadIdx.x] = (T)v;
    }
    if constexpr (std::is_same<T, _Float16>::value) rg_report(rgmax, ovf, 1u);
}

}  // namespace

extern "C" {

int mi355_cast16_fwd(const float* src, void* dst16, size_t n, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(src && dst16 && n > 0 && aligned16(src) && aligned16(dst16));
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    const long n8 = (long)(n / 8);
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 + 1 : 4096);
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* ovf = precision == MI355_PREC_FP16 ? mi355::range_word(st) : nullptr;
    MI355_TRACE(st, "cast16_kernel n=%zu", n);
    if (precision == MI355_PREC_FP16) cast16_kernel<_Float16><<<grid, 256, 0, st>>>(src, static_cast<_Float16*>(dst16), n8, (long)n, ovf);
    else                              cast16_kernel<__bf16><<<grid, 256, 0, st>>>(src, static_cast<__bf16*>(dst16), n8, (long)n, ovf);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

size_t mi355_linear16_workspace_bytes(int M, int N, int K) { return mi355::gemm16_p8_workspace_bytes(M, N, K); }

int mi355_linear16_fwd(const void* X16, const void* W16, const float* bias, const float* gamma, const float* resid, void* Y, int M,
                       int N, int K, int ldx, int ldy, int act, int out16, int precision, mi355_stream_t stream) {
    return mi355_linear16_ws_fwd(X16, W16, bias, gamma, resid, Y, M, N, K, ldx, ldy, act, out16, precision, nullptr, 0, stream);
}

// Y16 = act(T(X32) . W16^T + bias): the fp32 -> 16-bit cast of the activation rides in the X staging of gemm16_wslab (no 16-bit copy of X in HBM,
// one launch instead of mi355_cast16_fwd + mi355_linear16_fwd; the same bits).  MI355_EUNSUPPORTED -- nothing launched, no error text -- for
// shapes the slab-stationary kernel does not take or with "gemm_wslab" = 0: the caller casts and calls mi355_linear16_fwd.
int mi355_linear16_x32_fwd(const float* X32, const void* W16, const float* bias, void* Y16, int M, int N, int K, int ldx, int ldy, int act,
                           int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(X32 && W16 && Y16 && M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (!mi355::opt_gemm_wslab()) return MI355_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    G16Args g{};
    g.Af = X32; g.B = W16; g.C = Y16; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy; g.act = act;
    // validate BEFORE touching the range word: an unsupported shape must leave no trace (the caller's cast16 + linear16 then count as two producers)
    const int rc = mi355::gemm16_wslab_check(g, precision);
    if (rc != MI355_OK) return rc;
    if (precision == MI355_PREC_FP16) g.ovf = mi355::range_word(st);
    const int rc2 = mi355::gemm16_wslab(g, 1, precision, st);
    if (rc2 != MI355_OK) return rc2;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_linear16_ws_fwd(const void* X16, const void* W16, const float* bias, const float* gamma, const float* resid, void* Y, int M,
                          int N, int K, int ldx, int ldy, int act, int out16, int precision, void* ws, size_t ws_bytes,
                          mi355_stream_t stream) {
    MI355_CHECK_ARG(X16 && W16 && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((K % BK) || (N & 3) || (ldx & 7) || (ldy & 3) || !aligned16(X16) || !aligned16(W16) || !aligned16(Y) ||
        (bias && !aligned16(bias)) || (gamma && !aligned16(gamma)) || (resid && !aligned16(resid)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_fwd: needs K %% 64 == 0, N %% 4 == 0, 16-byte aligned rows (K=%d N=%d ldx=%d)",
                           K, N, ldx);
    G16Args g{};
    g.A = X16; g.B = W16; g.C = Y; g.bias = bias; g.gamma = gamma; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy; g.act = act;
    if (out16 && precision == MI355_PREC_FP16) g.ovf = mi355::range_word(static_cast<hipStream_t>(stream));    // fp16 range guard
    return mi355::linear16_dispatch(g, out16, precision, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_linear16_stats_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K, int ldx,
                             int ldy, int precision, float* row_stats, float eps, mi355_stream_t stream) {
    MI355_CHECK_ARG(X16 && W16 && resid && Y && row_stats && M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (!aligned16(X16) || !aligned16(W16) || !aligned16(Y) || !aligned16(resid) || (bias && !aligned16(bias)) || !mi355::opt_gemm_wreg())
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_stats_fwd: 16-byte aligned buffers and option gemm_wreg = 1 required");
    G16Args g{};
    g.A = X16; g.B = W16; g.C = Y; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy; g.act = MI355_ACT_NONE;
    g.row_stats = row_stats; g.ln_eps = eps;
    const int rc = mi355::gemm16_wreg(g, 0, precision, static_cast<hipStream_t>(stream));
    if (rc == MI355_EUNSUPPORTED)
        return mi355::fail(rc, "mi355_linear16_stats_fwd: built for N = K = 256 / 384, M >= 32 (got M=%d N=%d K=%d): use mi355_linear16_fwd and a statistics pass", M, N, K);
    if (rc != MI355_OK) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_linear16_ln16_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, const float* ln_w,
                            const float* ln_b, float eps, void* U16, int M, int N, int K, int ldx, int ldy, int ldu, int precision,
                            mi355_stream_t stream) {
    MI355_CHECK_ARG(X16 && W16 && resid && Y && ln_w && ln_b && U16 && M > 0 && N > 0 && K > 0 && ldx >= K && ldy >= N && ldu >= N);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (!aligned16(X16) || !aligned16(W16) || !aligned16(Y) || !aligned16(resid) || (bias && !aligned16(bias)) || !mi355::opt_gemm_wreg())
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_ln16_fwd: 16-byte aligned buffers and option gemm_wreg = 1 required");
    G16Args g{};
    g.A = X16; g.B = W16; g.C = Y; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy; g.act = MI355_ACT_NONE;
    g.ln16_out = U16; g.ln16_w = ln_w; g.ln16_b = ln_b; g.ln16_ld = ldu; g.ln_eps = eps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (precision == MI355_PREC_FP16) g.ovf = mi355::range_word(st);                // U16 is an fp16 operand tensor: a producer
    const int rc = mi355::gemm16_wreg(g, 0, precision, st);
    if (rc == MI355_EUNSUPPORTED)
        return mi355::fail(rc, "mi355_linear16_ln16_fwd: built for N = K = 256 / 384, M >= 32 (got M=%d N=%d K=%d): use mi355_linear16_fwd + mi355_layernorm16_fwd", M, N, K);
    if (rc != MI355_OK) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"

// Kernel choice of mi355_linear16_ws_fwd on a checked argument block (also the product of the patch embedding, gemm.hip, which
// brings a periodic residual table: only the persistent and the plain tile kernels read one).
int mi355::linear16_dispatch(const G16Args& g, int out16, int precision, void* ws, size_t ws_bytes, hipStream_t st) {
    const int M = g.M, N = g.N, K = g.K;
    long variant = mi355::opt_gemm_variant();
    if (g.resid_period) {
        if (K == 64 || K == 128 || out16 || !g.resid) return MI355_EUNSUPPORTED;
    }
    if (variant == 15) {                   // persistent 256 x 256 kernel (gemm16_p8.hip)
        const int rc = mi355::gemm16_p8(g, out16, precision, ws, ws_bytes, st);
        if (rc == MI355_EUNSUPPORTED) return mi355::fail(rc, "mi355_linear16_fwd: persistent kernel does not take this shape");
        if (rc != MI355_OK) return rc;
        MI355_LAUNCH_CHECK();
        return MI355_OK;
    }
    if (variant == 16) {                   // two-accumulator persistent 128 x 256 kernel (gemm16_pa.hip)
        const int rc = mi355::gemm16_pa(g, out16, precision, st);
        if (rc == MI355_EUNSUPPORTED) return mi355::fail(rc, "mi355_linear16_fwd: the two-accumulator kernel does not take this shape");
        if (rc != MI355_OK) return rc;
        MI355_LAUNCH_CHECK();
        return MI355_OK;
    }
    if (variant == 17) {                   // one-wave-per-SIMD persistent 256 x 256 kernel (gemm16_w4.hip)
        const int rc = mi355::gemm16_w4(g, out16, precision, st);
        if (rc == MI355_EUNSUPPORTED) return mi355::fail(rc, "mi355_linear16_fwd: the one-wave-per-SIMD kernel does not take this shape");
        if (rc != MI355_OK) return rc;
        MI355_LAUNCH_CHECK();
        return MI355_OK;
    }
    if (variant == 0 && out16 && K == 768 && (mi355::opt_gemm_wst() == 2 || mi355::opt_gemm_wst() == 4 || ((mi355::opt_gemm_wst() & 1) && g.act == MI355_ACT_NONE))) {
        // ViT qkv / fc1, opt-in: a 192-column slab of W stays in the registers of a workgroup, X streams through LDS once per slab (gemm16_wst.hip)
        const int rc = mi355::gemm16_wst(g, out16, precision, st);
        if (rc == MI355_OK) {
            MI355_LAUNCH_CHECK();
            return MI355_OK;
        }
        if (rc != MI355_EUNSUPPORTED) return rc;
    }
    if (variant == 0 && (K == 64 || K == 128) && M >= 2048 && (!out16 || (N & 7) == 0)) {       // short-K, HBM-bound: weight-stationary streaming kernel
        int ncu = 256, dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const int bn = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
        const int tiles_n = cdiv(N, bn), tiles_m = cdiv(M, 128);
        const int per_cu = (K == 64 && bn <= 256) ? 2 : 1;
        int workers = (ncu * per_cu) / tiles_n;
        if (workers < 1) workers = 1;
        if (workers > tiles_m) workers = tiles_m;
        const int grid = tiles_n * workers;
        MI355_TRACE(st, "gemm16_ws_kernel<%s> M=%d N=%d K=%d", out16 ? "out16" : "out32", M, N, K);
#define WS(T_, O_, BN_, KK_) gemm16_ws_kernel<T_, O_, BN_, KK_><<<grid, 512, 0, st>>>(g, workers)
#define WS_BY_SHAPE(T_, O_)                                                   \
        do {                                                                  \
            if (K == 64) { if (bn == 256) WS(T_, O_, 256, 64); else if (bn == 128) WS(T_, O_, 128, 64); else WS(T_, O_, 64, 64); }    \
            else         { if (bn == 256) WS(T_, O_, 256, 128); else if (bn == 128) WS(T_, O_, 128, 128); else WS(T_, O_, 64, 128); } \
        } while (0)
        if (precision == MI355_PREC_FP16) { if (out16) WS_BY_SHAPE(_Float16, true); else WS_BY_SHAPE(_Float16, false); }
        else                              { if (out16) WS_BY_SHAPE(__bf16, true); else WS_BY_SHAPE(__bf16, false); }
#undef WS_BY_SHAPE
#undef WS
        MI355_LAUNCH_CHECK();
        return MI355_OK;
    }
    if (variant == 0 && out16 && (K == 256 || K == 384 || K == 512) && !g.gamma && !g.resid &&
        (mi355::opt_gemm_wslab() == 2 || (mi355::opt_gemm_wslab() == 1 && (g.act == MI355_ACT_GELU || (M & 255))))) {
        // short reductions, 16-bit output (XCiT / CSWin stage 3-4 / Mixer qkv and fc1): a column slab of W stationary in registers, X through LDS
        // once per slab, 32-row tiles (gemm16_wslab.hip; bit-identical to the tile kernels).  Default policy: where it measured faster -- GELU
        // epilogues (5-12 %) and row counts off the 256-row grid (which otherwise fall to gemm16_p8: 88-96 -> 65-69 us); plain epilogues are a tie.
        const int rc = mi355::gemm16_wslab(g, out16, precision, st);
        if (rc == MI355_OK) {
            MI355_LAUNCH_CHECK();
            return MI355_OK;
        }
        if (rc != MI355_EUNSUPPORTED) return rc;
    }
    if (variant == 0 && out16 && K >= 256 && !g.gamma && !g.resid && mi355::opt_gemm_pa() &&
        (K <= 512 || (mi355::opt_gemm_pa16() >= 1 && g.act == MI355_ACT_GELU) || mi355::opt_gemm_pa16() >= 2)) {
        // 16-bit outputs with a SHORT reduction (4 .. 8 K-tiles: CSWin stage 3 / 4, XCiT, the Mixer's token mixing): on the persistent
        // 256 x 256 kernel the epilogue of such a tile (bias / GELU / convert / store, nothing to overlap it with) is as long as its
        // main loop; the two-accumulator kernel packs two or three convert pieces into every barrier interval of the next tile's
        // main loop.  Round 4, same box, same process: CSWin s3 qkv 40 -> 32 us, fc1 74 -> 60, XCiT qkv 70 -> 54, fc1 106 -> 89,
        // Mixer fc1 167 -> 136; bit-identical results.  Longer reductions (K >= 576): the 256 x 256 kernel's main
        // loop stages 1.5 x fewer bytes per flop, but its GELU epilogue is fully exposed (55 us of ViT-Base's fc1) where the pieces of
        // this kernel hide most of it (29 us): option "gemm_pa16" = 1 (default) sends GELU epilogues here (fc1 313 -> 291 us in a
        // same-process A/B), 2 every 16-bit output (qkv: 221 -> 207 on one kind of box, 197 -> 208 on the other: not the default), 0 neither.
        const int ncu = mi355::resident_slots(1);
        const long tiles = (N & 255) ? (long)(M / 256) * (N / 128) : (long)cdiv(M, 128) * (N / 256);
        if (2 * tiles >= ncu) {
            const int rc = mi355::gemm16_pa(g, out16, precision, st);
            if (rc == MI355_OK) {
                MI355_LAUNCH_CHECK();
                return MI355_OK;
            }
            if (rc != MI355_EUNSUPPORTED) return rc;
        }
    }
    if (variant == 0 && !out16 && N == K && (K == 256 || K == 384) && mi355::opt_gemm_wreg()) {
        // square short products with an fp32 (+ residual) output (XCiT proj 384 x 384, CSWin stage-3 proj 256 x 256): bound by the residual /
        // output stream, not by the matrix pipes -- weights stationary in registers, X / residual / Y each cross HBM once (gemm16_wreg.hip;
        // round 6: XCiT proj 70 -> see profiles/r06_gemm_wreg.md; bit-identical to the tile kernels)
        const int rc = mi355::gemm16_wreg(g, out16, precision, st);
        if (rc == MI355_OK) {
            MI355_LAUNCH_CHECK();
            return MI355_OK;
        }
        if (rc != MI355_EUNSUPPORTED) return rc;
    }
    if (variant == 0 && !out16 && (g.resid || N <= 768) && mi355::opt_gemm_pa()) {
        // fp32 (+ residual) outputs: the two-accumulator persistent kernel (gemm16_pa.hip) hides the residual / store round trips of
        // tile i under the main loop of tile i + 1 (ViT-Base proj 0.130 -> 0.106 ms, fc2 0.266 -> 0.259; profiles/r03_gemm_pa.md).
        // No inter-workgroup exchange: safe under hipGraph capture, and a row's bits never depend on where its tile falls.
        // Round 4: widths that are a multiple of 128 but not of 256 (XCiT: N = 384) run 256 x 128 tiles (operand roles swapped), and
        // half a round of tiles is enough (CSWin stage 4 fc2: 196 tiles, 41 -> 36 us).
        const int ncu = mi355::resident_slots(1);
        const long tiles = (N & 255) ? (long)(M / 256) * (N / 128) : (long)cdiv(M, 128) * (N / 256);
        if (2 * tiles >= ncu) {
            // A last round that is nearly empty costs a whole tile time on a persistent kernel (the Mixer's fc2: 784 tiles = 3.06
            // rounds, XCiT's fc2: 588 = 2.30).  With a long reduction that is tens of microseconds, so the two-accumulator kernel
            // gets the rows of its whole rounds and the rest goes, as a second launch, to ring-pipelined small tiles that cover all
            // CUs (option "gemm_pa_tail").  Both kernels add a row's K-tiles in the same order: bit-identical to the unsplit launch.
            const int bmt = (N & 255) ? 256 : 128, tn = (N & 255) ? N / 128 : N / 256;
            const long left = tiles % ncu, pct = mi355::opt_gemm_pa_tail();
            if (pct > 0 && tiles > ncu && left > 0 && left * 100 <= pct * ncu && K >= 1024 && M % bmt == 0 && !g.resid_period && !g.lnc_a) {
                const int rows1 = (int)((tiles - left) / tn) * bmt, rows2 = M - rows1;
                G16Args g1 = g, g2 = g;
                g1.M = rows1;
                g2.M = rows2;
                g2.A = static_cast<const char*>(g.A) + (size_t)rows1 * g.lda * 2;
                g2.C = static_cast<char*>(g.C) + (size_t)rows1 * g.ldc * 4;
                if (g.resid) g2.resid = g.resid + (size_t)rows1 * g.ldc;
                const int rc = mi355::gemm16_pa(g1, out16, precision, st);
                if (rc == MI355_OK) {
                    MI355_LAUNCH_CHECK();
                    // 32 x 64 tiles, four waves, a ring of ten K-tiles (120 KB): 1 024 x 512 left-over outputs = one workgroup per CU.  The
                    // small tile stages 12 KB per 0.26 MFLOP through an LDS-DMA path that tops out near 27 B/clk/CU, so
                    // it pays for FEW left-over tiles only (round 4: Mixer fc2, 16 of 784 tiles, 143 -> 135 us and 490 -> 475 us for the
                    // block; XCiT fc2, 76 of 588, 92 -> 102 us with 64 x 128 tiles: hence the 10 % default).
                    const int grid = cdiv(rows2, 32) * cdiv(N, 64);
                    MI355_TRACE(st, "gemm16_kernel<tail 32x64> M=%d N=%d K=%d", rows2, N, K);
                    if (precision == MI355_PREC_FP16) gemm16_kernel<_Float16, false, 32, 64, 1, 4, false, 10><<<grid, 256, 0, st>>>(g2);
                    else                              gemm16_kernel<__bf16, false, 32, 64, 1, 4, false, 10><<<grid, 256, 0, st>>>(g2);
                    MI355_LAUNCH_CHECK();
                    return MI355_OK;
                }
                if (rc != MI355_EUNSUPPORTED) return rc;
            }
            const int rc = mi355::gemm16_pa(g, out16, precision, st);
            if (rc == MI355_OK) {
                MI355_LAUNCH_CHECK();
                return MI355_OK;
            }
            if (rc != MI355_EUNSUPPORTED) return rc;
        }
    }
    if (variant == 0 && (N & 7) == 0 && !(out16 && g.resid)) {
        // Persistent 256 x 256 kernel (gemm16_p8.hip) for every shape with at least one full round of tiles (one workgroup per CU)
        // and K >= 256: it is >= the best 3-workgroups-per-CU variant on all 17 shapes of profiles/r02_gemm_p8.md except CSWin s3 fc1
        // (inside the run-to-run band), shapes just above a round boundary included (the last partial round costs a whole tile time
        // either way; long reductions cut it along K).
        const int ncu = mi355::resident_slots(1);
        const long ntiles = (long)cdiv(M, 256) * cdiv(N, 256);
        if (ntiles >= ncu && K >= 256) {
            // 16-bit outputs of a mid-length reduction (ViT qkv: K = 768): one wave per SIMD with 128 x 128 outputs each needs a third fewer LDS
            // bytes per flop and, with its two-slab pipelined epilogue, 5.8 k instead of ~11 k exposed cycles per tile (round 5, same process:
            // qkv 0.193-0.195 -> 0.179-0.182 ms).  Long reductions keep the split last round of the eight-wave kernel.
            if (out16 && K >= 576 && K < 1536 && mi355::opt_gemm_w4()) {
                const int rc = mi355::gemm16_w4(g, out16, precision, st);
                if (rc == MI355_OK) {
                    MI355_LAUNCH_CHECK();
                    return MI355_OK;
                }
                if (rc != MI355_EUNSUPPORTED) return rc;
            }
            const int rc = mi355::gemm16_p8(g, out16, precision, ws, ws_bytes, st);
            if (rc == MI355_OK) {
                MI355_LAUNCH_CHECK();
                return MI355_OK;
            }
            if (rc != MI355_EUNSUPPORTED) return rc;                 // e.g. MI355_ESYNC: an earlier launch failed, reported once
        }
    }
    if (variant == 0) {        // default (profiles/r01_gemm_variants.md): 8 waves on a 128x256 tile, single LDS buffer, 3 workgroups
        variant = (N <= 64) ? 9 : (N < 256 ? 1 : 7);   // per CU; narrow outputs use 256x64 / 128x128 tiles instead
        int ncu = 256, dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        // fewer 128x256 tiles than CUs (CSWin stage 4: M = 12544, N = 512 -> 196 tiles): halve the tile so that the chip fills
        // (measured 57 -> 43 us on the K = 2048 fc2 of that stage, 25 -> 22 us on its proj)
        if (variant == 7 && (long)cdiv(M, 128) * cdiv(N, 256) < ncu) variant = 1;
    }
    else if (variant == 8) variant = 0;     // 8 = plain 128x128 without priority hints (tuning experiments)
    MI355_TRACE(st, "gemm16_kernel<variant %ld,%s> M=%d N=%d K=%d%s", variant, out16 ? "out16" : "out32", M, N, K, g.act == MI355_ACT_GELU ? " gelu" : "");
#define LAUNCH(T_, O_, BM_, BN_, WM_, WN_, P_, S_)                                                         \
    gemm16_kernel<T_, O_, BM_, BN_, WM_, WN_, P_, S_><<<cdiv(M, BM_) * cdiv(N, BN_), WM_ * WN_ * 64, 0, st>>>(g)
#define BY_VARIANT(T_, O_)                                                 \
    do {                                                                   \
        switch (variant) {                                                 \
            case 1: LAUNCH(T_, O_, 128, 128, 2, 2, true, 2); break;        \
            case 2: LAUNCH(T_, O_, 256, 128, 4, 2, false, 2); break;       \
            case 3: LAUNCH(T_, O_, 256, 256, 2, 4, false, 2); break;       \
            case 4: LAUNCH(T_, O_, 256, 256, 2, 4, true, 2); break;        \
            case 5: LAUNCH(T_, O_, 128, 128, 2, 2, true, 1); break;        \
            case 6: LAUNCH(T_, O_, 256, 128, 4, 2, true, 1); break;        \
            case 7: LAUNCH(T_, O_, 128, 256, 2, 4, true, 1); break;        \
            case 9: LAUNCH(T_, O_, 256, 64, 4, 1, true, 1); break;         \
            default: LAUNCH(T_, O_, 128, 128, 2, 2, false, 2); break;      \
        }                                                                  \
    } while (0)
    if (precision == MI355_PREC_FP16) {
        if (out16) BY_VARIANT(_Float16, true); else BY_VARIANT(_Float16, false);
    } else {
        if (out16) BY_VARIANT(__bf16, true); else BY_VARIANT(__bf16, false);
    }
#undef BY_VARIANT
#undef LAUNCH
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

extern "C" {

int mi355_linear16_tr_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K,
                          int ldx, int rows_per_image, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(X16 && W16 && Y && M > 0 && N > 0 && K > 0 && ldx >= K && rows_per_image > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if ((K % BK) || (ldx & 7) || (rows_per_image & 3) || (M % rows_per_image) || !aligned16(X16) || !aligned16(W16) || !aligned16(Y) ||
        (resid && !aligned16(resid)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_linear16_tr_fwd: needs K %% 64 == 0, rows_per_image %% 4 == 0 dividing M (K=%d M=%d rows=%d)",
                           K, M, rows_per_image);
    G16Args g{};
    g.A = X16; g.B = W16; g.C = Y; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = N; g.act = MI355_ACT_NONE; g.tr_rows = rows_per_image;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool wide = N > 128 && ((N & 255) == 0 || (N & 127) != 0);    // N = 384: three exact 128-wide tiles instead of 256 + half-empty 256
    MI355_TRACE(st, "gemm16_kernel<transposed out> M=%d N=%d K=%d", M, N, K);
#define TRL(T_, BN_, WN_) gemm16_kernel<T_, false, 128, BN_, 2, WN_, true, 1, true><<<cdiv(M, 128) * cdiv(N, BN_), 2 * WN_ * 64, 0, st>>>(g)
    if (precision == MI355_PREC_FP16) { if (wide) TRL(_Float16, 256, 4); else TRL(_Float16, 128, 2); }
    else                              { if (wide) TRL(__bf16, 256, 4); else TRL(__bf16, 128, 2); }
#undef TRL
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_ln_linear16_fwd(const float* X, const void* W16, const float* bias, void* Y, int M, int N, int K, int ldx, int ldy, float eps,
                          int act, int out16, int precision, mi355_stream_t stream) {
    MI355_CHECK_ARG(X && W16 && Y && M > 0 && N > 0 && ldx >= K && ldy >= N);
    MI355_CHECK_ARG(act == MI355_ACT_NONE || act == MI355_ACT_GELU);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    if (!(K == 64 || K == 128) || (N & 7) || (ldx & 3) || (ldy & 7) || !aligned16(X) || !aligned16(W16) || !aligned16(Y) || (bias && !aligned16(bias)))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_ln_linear16_fwd: built for K = 64 / 128, N %% 8 == 0, 16-byte aligned rows (K=%d N=%d)", K, N);
    G16Args g{};
    g.Af = X; g.B = W16; g.C = Y; g.bias = bias; g.ln_eps = eps;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy; g.act = act;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out16 && precision == MI355_PREC_FP16) g.ovf = mi355::range_word(st);
    int ncu = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int bn = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const int tiles_n = cdiv(N, bn), tiles_m = cdiv(M, 128);
    int workers = (ncu * 2) / tiles_n;
    if (workers < 1) workers = 1;
    if (workers > tiles_m) workers = tiles_m;
    const int grid = tiles_n * workers;
#define WSL(T_, O_, BN_, KK_) gemm16_ws_kernel<T_, O_, BN_, KK_, true><<<grid, 512, 0, st>>>(g, workers)
#define WSL_BY_SHAPE(T_, O_)                                                   \
    do {                                                                       \
        if (K == 64) { if (bn == 256) WSL(T_, O_, 256, 64); else if (bn == 128) WSL(T_, O_, 128, 64); else WSL(T_, O_, 64, 64); }    \
        else         { if (bn == 256) WSL(T_, O_, 256, 128); else if (bn == 128) WSL(T_, O_, 128, 128); else WSL(T_, O_, 64, 128); } \
    } while (0)
    if (precision == MI355_PREC_FP16) { if (out16) WSL_BY_SHAPE(_Float16, true); else WSL_BY_SHAPE(_Float16, false); }
    else                              { if (out16) WSL_BY_SHAPE(__bf16, true); else WSL_BY_SHAPE(__bf16, false); }
#undef WSL_BY_SHAPE
#undef WSL
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
