// chan_fused.hip -- SINGLE-PASS SELayer / ECALayer for gfx950: x is read from HBM once and y written once.
//
// The multi-pass path (chan_attn.hip) must stream x twice because the gate of an image depends on the whole image.  Here
// a workgroup keeps its slice of the image -- 4 channels, one per wave, HW floats each (12.5 KB at 56x56) -- IN REGISTERS
// between pooling and scaling, and the 4-channel means are exchanged between the C/4 sibling workgroups of the image
// through HBM-side scratch with the placement-independent publish/consume protocol of cdna_hip_programming.md G16:
//
//   slice ticket  = atomicAdd(ticket)            dynamic hand-out in image order (no residency assumption: progress only
//                                                needs >= C/4 workgroups running)
//   load slice    -> registers (NV float4 per lane)
//   publish       : lane 0 of each wave stores its channel mean write-through (sc1, agent-scope relaxed atomic store),
//                   every storing wave drains vmcnt(0), __syncthreads(), one lane bumps arrive[image] (agent relaxed add)
//   consume       : one lane polls arrive[image] (relaxed, s_sleep) until C/4 arrivals, __syncthreads(), every thread
//                   reads the image's means with sc1 loads (they bypass L1; valid because the producers stored sc1)
//   gate + scale  : the 4 gates are recomputed from the means (2-layer excite / k-tap conv), y = x * g from registers,
//                   non-temporal stores.
// State (ticket, arrive[]) is zeroed by a memset node in front of every launch; every spin is bounded (error word).
//
// STATUS: correct (tests/test_chan_attn_gpu.py::test_single_pass_*) but NOT the default.  Measured on MI355X at the C2 shape:
// 1.5 ms vs 0.40 ms for the two-pass path.  Each slice pays a chain of dependent fabric round trips under streaming load
// (write-through mean -> arrival atomic -> poll -> mean loads, ~5 us each: MI355X_MICROARCH "handoff" rows) while the
// register file can only keep ~4 slices (200 KB) per CU in flight: Little's law caps it near 1-2 TB/s.  Kept behind
// mi355_set_option("fused", 1|2) as the measured negative result.
#include "common.h"

namespace {

using v4f = float __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

#define AGENT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct FusedArgs {
    const float* x; float* y; const float* wa; const float* wb;
    float* means; u32* arrive; u32* ticket; u32* err;
    int B, C, Cr, HW, n4, spi, total;
};

constexpr int CPW = 4;                 // channels per workgroup (= waves)
constexpr u32 SPIN_LIMIT = 1u << 22;   // ~1 s of polling before giving up with an error word

template <int MODE, int NV>
__global__ __launch_bounds__(256) void se_eca_fused_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // p[C] | h[Cr] | g[CPW]
    __shared__ u32 s_tk;
    float* s_p = smem;
    float* s_h = smem + a.C;
    float* s_g = s_h + (MODE == 0 ? a.Cr : 0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (;;) {
        if (t == 0) s_tk = __hip_atomic_fetch_add(a.ticket, 1u, AGENT_RLX);
        __syncthreads();
        const u32 tk = s_tk;
        __syncthreads();                                             // s_tk may be rewritten next iteration
        if (tk >= (u32)a.total) return;
        const int b = tk / a.spi, c0 = (tk % a.spi) * CPW;
        const int c = c0 + wave;
        const long off = ((long)b * a.C + c) * a.HW;

        // ---- load the channel row into registers, pool it -----------------------------------------------------------
        v4f r[NV];
        const v4f* xr = reinterpret_cast<const v4f*>(a.x + off);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            r[j] = (i < a.n4) ? __builtin_nontemporal_load(&xr[i]) : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) { s0 += r[j].x; s1 += r[j].y; s2 += r[j].z; s3 += r[j].w; }
        const float mean = wave_sum((s0 + s1) + (s2 + s3)) / (float)a.HW;

        // ---- publish (write-through payload, drained, one arrival per workgroup) ---------------------------------------
        if (lane == 0) __hip_atomic_store(a.means + (long)b * a.C + c, mean, AGENT_RLX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(a.arrive + b, 1u, AGENT_RLX);
            // ---- consume: poll until every sibling of this image has arrived ----------------------------------------
            u32 spins = 0;
            while (__hip_atomic_load(a.arrive + b, AGENT_RLX) < (u32)a.spi) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SPIN_LIMIT) { __hip_atomic_store(a.err, 1u, AGENT_RLX); break; }
            }
        }
        __syncthreads();

        // ---- gates of this workgroup's channels from the image's means (sc1 loads) ---------------------------------------
        const float* mb = a.means + (long)b * a.C;
        if constexpr (MODE == 0) {
            for (int cc = t; cc < a.C; cc += 256) s_p[cc] = __hip_atomic_load(mb + cc, AGENT_RLX);
            __syncthreads();
            const int part = t & 15, jl = t >> 4;
            for (int j0 = 0; j0 < a.Cr; j0 += 16) {
                const int j = j0 + jl;
                float acc = 0.f;
                if (j < a.Cr) {
                    const float* wrow = a.wa + (long)j * a.C;
                    for (int cc = part; cc < a.C; cc += 16) acc += wrow[cc] * s_p[cc];
                }
                acc += __shfl_xor(acc, 8, WAVE);
                acc += __shfl_xor(acc, 4, WAVE);
                acc += __shfl_xor(acc, 2, WAVE);
                acc += __shfl_xor(acc, 1, WAVE);
                if (part == 0 && j < a.Cr) s_h[j] = fmaxf(acc, 0.f);
            }
            __syncthreads();
            if (t < CPW) {
                const float* w2r = a.wb + (long)(c0 + t) * a.Cr;
                float z = 0.f;
                for (int j = 0; j < a.Cr; ++j) z += w2r[j] * s_h[j];
                s_g[t] = sigmoidf_(z);
            }
        } else {
            if (t < CPW) {
                const int k = a.Cr, pad = (k - 1) / 2, ch = c0 + t;
                float z = 0.f;
                for (int j = 0; j < k; ++j) {
                    const int cc = ch + j - pad;
                    if (cc >= 0 && cc < a.C) z += a.wa[j] * __hip_atomic_load(mb + cc, AGENT_RLX);
                }
                s_g[t] = sigmoidf_(z);
            }
        }
        __syncthreads();

        // ---- scale from registers, stream out ---------------------------------------------------------------------------
        const float g = s_g[wave];
        v4f* yr = reinterpret_cast<v4f*>(a.y + off);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < a.n4) __builtin_nontemporal_store(r[j] * g, &yr[i]);
        }
        __syncthreads();                                             // s_g / s_p reuse
    }
}

template <int MODE>
hipError_t launch_fused(const FusedArgs& a, int nv, size_t smem, hipStream_t st) {
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
#define GO(NV_)                                                                                                     \
    do {                                                                                                            \
        static int occ_cache = 0;                    /* per instantiation; smem varies little, query once */        \
        int occ = occ_cache;                                                                                        \
        if (occ == 0) {                                                                                             \
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, se_eca_fused_kernel<MODE, NV_>, 256, smem); \
            if (e != hipSuccess) return e;                                                                          \
            if (occ < 1) occ = 1;                                                                                   \
            if (occ > 8) occ = 8;                                                                                   \
            occ_cache = occ;                                                                                        \
        }                                                                                                           \
        long grid = (long)ncu * occ;                                                                                \
        if (grid > a.total) grid = a.total;                                                                         \
        se_eca_fused_kernel<MODE, NV_><<<(int)grid, 256, smem, st>>>(a);                                           \
    } while (0)
    if (nv <= 1) GO(1);
    else if (nv <= 2) GO(2);
    else if (nv <= 4) GO(4);
    else if (nv <= 8) GO(8);
    else if (nv <= 13) GO(13);
    else GO(16);
#undef GO
    return hipGetLastError();
}

}  // namespace

namespace mi355 {

// extra workspace (beyond the pooled means) needed by the single-pass path: arrive[B] | ticket | err   (u32 each)
size_t fused_state_bytes(int B) { return (((size_t)B + 2) * sizeof(u32) + 15) & ~(size_t)15; }

// true when the single-pass kernel handles this shape (the caller falls back to the multi-pass path otherwise)
bool fused_applicable(int B, int C, int H, int W) {
    const long HW = (long)H * W;
    const bool shape_ok = (HW % 4 == 0) && (HW / 4 <= 16 * 64) && (C % CPW == 0) && (C <= 4096);
    if (opt_fused() >= 2) return shape_ok;                 // forced (tests)
    return shape_ok && ((long)B * C * HW * 4 >= (64L << 20));   // small problems: sync latency dominates, keep 2 launches
}

// mode 0 = SE (wa = W1 (Cr,C), wb = W2 (C,Cr)); mode 1 = ECA (wa = taps, Cr = k).  `means` (B*C floats) and `state`
// (fused_state_bytes) live in the caller's workspace.
int se_eca_fused(int mode, const float* x, const float* wa, const float* wb, float* y, int B, int C, int Cr, int H, int W,
                 float* means, void* state, hipStream_t st) {
    FusedArgs a{};
    a.x = x; a.y = y; a.wa = wa; a.wb = wb; a.means = means;
    a.arrive = static_cast<u32*>(state);
    a.ticket = a.arrive + B;
    a.err = a.ticket + 1;
    a.B = B; a.C = C; a.Cr = Cr; a.HW = H * W; a.n4 = a.HW / 4; a.spi = C / CPW; a.total = B * a.spi;
    hipError_t e = hipMemsetAsync(state, 0, ((size_t)B + 2) * sizeof(u32), st);
    if (e != hipSuccess) return fail(MI355_EHIP, "se_eca_fused: memset -> %s", hipGetErrorString(e));
    const size_t smem = (size_t)(C + (mode == 0 ? Cr : 0) + CPW) * sizeof(float);
    const int nv = (a.n4 + 63) / 64;
    e = (mode == 0) ? launch_fused<0>(a, nv, smem, st) : launch_fused<1>(a, nv, smem, st);
    if (e != hipSuccess) return fail(MI355_EHIP, "se_eca_fused: launch -> %s", hipGetErrorString(e));
    return MI355_OK;
}

}  // namespace mi355
