// chan_fused.hip -- single-read SELayer / ECALayer for gfx950: x leaves HBM once, y is written once.
//
// The multi-pass path (chan_attn.hip) streams x twice because the gate of an image depends on the whole image.  Here a 512-thread
// workgroup keeps 8 channel rows (one per wave, HW floats each: 12.5 KB at 56x56) IN REGISTERS between pooling and scaling.
//   ECA  (eca_halo_kernel):   the k-tap gate only needs the neighbouring channels: the workgroup re-sums the k-1 halo rows next to its
//                             slab instead of talking to anybody.
//   SE   (se_single_kernel):  the gate needs every channel mean of the image: each wave publishes its mean as one 8-byte {mean, tag}
//                             granule (write-through store), every workgroup of the image sweeps the image's granules.
// Round 1's first attempt at this (arrival counter + flag, four dependent fabric round trips per slice) measured 1.5 ms against
// 0.40 ms for two passes and is gone; DESIGN.md 6.1 keeps the numbers.
#include "common.h"
#include "bufops.h"

namespace {

using v4f = float __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

#define AGENT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// ---- ECA without any cross-workgroup exchange ------------------------------------------------------------------------------
// The ECA gate of channel c only needs the means of channels c-pad..c+pad (eca.py:26-30, k taps, zero padding).  A workgroup
// therefore keeps ECW = 8 channel rows in registers (one per wave) and additionally SUMS the 2*pad halo rows next to its slab
// (streamed, not kept).  Slices are numbered so that the workgroups an XCD receives (blockIdx % 8) own consecutive channel
// groups: the halo rows of one workgroup are the resident rows of its neighbours on the same XCD at the same moment, so the
// second request for a row is served by that XCD's L2 and the fabric/HBM traffic stays at one read + one write of x.
// Every mean -- own row or halo row -- is accumulated by ONE wave in the same lane/step order, so the value of mean(b,c) does
// not depend on which workgroup computes it.
constexpr int ECW = 8;

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool NTS>
__global__ __launch_bounds__(512, NV <= 13 ? 6 : 4) void eca_halo_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                          float* __restrict__ y, int C, int k, int HW, int gpi, int total,
                                                          int per_xcd) {
    __shared__ float s_mean[ECW + 8];                        // means of channels c0-pad .. c0+ECW+pad-1
    const int s = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (s >= total) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pad = (k - 1) >> 1;
    const int b = s / gpi, c0 = (s - b * gpi) * ECW;
    const float inv = 1.0f / (float)HW;
    const float* img = x + (long)b * C * HW;
    // rows through buffer descriptors: lanes beyond the row read zeros and their stores are dropped, so the streaming loops carry
    // no predicates and one VGPR of address state (<= 80 VGPRs: three workgroups per CU)
    const u32 cw = __builtin_amdgcn_readfirstlane((u32)(c0 + wave));
    const rsrc_t rx = make_rsrc(img + (long)cw * HW, (u32)HW * 4u);
    const u32 voff = (u32)lane * 16u;
    v4f r[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (u32)j * 1024u, 0));
    // halo row of this wave (if any): same lane/step order as an own row, streamed
    int hc = -1, hslot = 0;
    if (wave < 2 * pad) {
        hc = (wave < pad) ? c0 - pad + wave : c0 + ECW + (wave - pad);
        hslot = (wave < pad) ? wave : ECW + wave;
    }
    const bool halo_live = hc >= 0 && hc < C;                 // wave-uniform
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
    if (halo_live) {
        const u32 hcw = __builtin_amdgcn_readfirstlane((u32)hc);
        const rsrc_t rh = make_rsrc(img + (long)hcw * HW, (u32)HW * 4u);
#pragma unroll 4
        for (int j = 0; j < NV; ++j) {                        // a real loop (4 loads in flight): the own row already fills the registers
            const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rh, voff, (u32)j * 1024u, 0));
            h0 += v.x; h1 += v.y; h2 += v.z; h3 += v.w;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { s0 += r[j].x; s1 += r[j].y; s2 += r[j].z; s3 += r[j].w; }
    const float mean = wave_sum_sw((s0 + s1) + (s2 + s3)) * inv;
    if (lane == 0) s_mean[pad + wave] = mean;
    if (wave < 2 * pad) {
        const float hm = halo_live ? wave_sum_sw((h0 + h1) + (h2 + h3)) * inv : 0.f;
        if (lane == 0) s_mean[hslot] = hm;
    }
    __syncthreads();
    float z = 0.f;
    for (int j = 0; j < k; ++j) z += taps[j] * s_mean[wave + j];
    const float g = sigmoidf_(z);
    const rsrc_t ry = make_rsrc(y + ((long)b * C + cw) * HW, (u32)HW * 4u);
    u32 ob = voff;                                            // the row step rides in the VGPR offset of the stores: cbam_single.hip
    asm volatile("" : "+v"(ob));
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const v4f o = r[j] * g;
        if (NTS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, AUX_NT);
        else     __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, 0);
    }
}

// ---- SE, x read once: register-resident rows + data-tagged granules --------------------------------------------------------------
// Same geometry as the ECA kernel above (8 waves = 8 channel rows held in registers, two workgroups per CU), but the SE gate
// needs the means of ALL channels of the image.  Each wave publishes its mean as ONE naturally aligned 8-byte {mean, tag}
// granule with a single sc1 (write-through) store -- data and "ready" travel together, so there is no flag, no drain and no
// arrival counter (MI355X_MICROARCH.md price list: handoff-1to1 / allgather rows).  Every workgroup of the image then sweeps
// the image's C granules with sc1 loads (one per thread) until all tags match, computes the excitation MLP redundantly
// (C*Cr MACs) and scales its rows from registers.  Slices are handed out in image order by a ticket that is prefetched one
// slice ahead, so progress needs only C/8 running workgroups and never a particular placement.  The tag of a launch is `epoch + 1`,
// read from the workspace; the workgroup that draws the LAST ticket of a launch (every workgroup draws one ticket per slice plus one
// that tells it to stop: total + gridDim.x draws) sets the ticket word back to zero and advances the epoch.  Nothing about a launch
// lives on the host, so eager launches and hipGraph replays can be mixed freely on one workspace, and the granule array is only
// zeroed when the history of the workspace is unknown.  Polls are bounded.
typedef unsigned long long u64;

struct SeSingleArgs {
    const float* x; float* y; const float* w1; const float* w2; const float* b1; const float* b2;
    u64* gran; u32* ticket; u32* epoch; u32* err; u32* herr;   // err: workspace word (debug), herr: pinned host word every later call checks
    u32 spin;
    int gate;
    int C, Cr, HW, n4, gpi, total;
    float inv;                  // 1.0f / HW, divided on the host (IEEE division either way: the same bits the kernel used to compute)
};

// EXTRA: the SE variants of the reference's CNNs (biases of the two excitation layers, hard-sigmoid gate: SeExtra).  The plain SELayer
// instantiation carries none of their pointers, null checks and select masks -- scalar registers the 80-VGPR build cannot spare (past
// 102 SGPRs hipcc parks uniform values in VGPRs and then spills those).
template <int NV, bool NTS, bool WLDS, int OCC, bool EXTRA>
__global__ __launch_bounds__(512, 2 * OCC) void se_single_kernel(const SeSingleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // p[C] | h[Cr] | (WLDS: W1[Cr*C] | W2[C*Cr])
    __shared__ u32 s_tk[2];
    __shared__ u32 s_ep;
    __shared__ u32 s_ok[2][8];                                        // per-wave "all my granules are in" votes, double-buffered by sweep parity
    float* s_p = smem;
    float* s_h = smem + a.C;
    float* s_w1 = s_h + a.Cr;
    float* s_w2 = s_w1 + a.Cr * a.C;
    const int t0 = threadIdx.x;
    const float inv = a.inv;
    // one ticket; the last draw of the launch (number total + gridDim.x - 1) resets the ticket word and advances the epoch
    const u32 last_draw = (u32)a.total + gridDim.x - 1u;
    auto draw = [&](u32 ep) -> u32 {
        const u32 v = __hip_atomic_fetch_add(a.ticket, 1u, AGENT_RLX);
        if (v == last_draw) {
            u32 e1;                                                    // formed HERE: a hoisted VGPR copy of epoch + 1 lived (spilled) for the whole kernel
            asm volatile("v_mov_b32 %0, %1" : "=v"(e1) : "s"(ep + 1u));
            __hip_atomic_store(a.ticket, 0u, AGENT_RLX);
            __hip_atomic_store(a.epoch, e1, AGENT_RLX);
        }
        return v;
    };
    if (t0 == 0) {
        // read BEFORE the first draw (acquire: the fetch_add below may not be performed ahead of this load -- they are different
        // addresses, a relaxed pair has no order): the epoch cannot move until this workgroup has drawn its stop ticket
        const u32 ep = __hip_atomic_load(a.epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        s_ep = ep;
        s_tk[0] = draw(ep);
    }
    if (WLDS) {                                                       // both weight matrices stay in LDS for every slice
        const int nw = a.Cr * a.C;
        for (int i = t0; i < nw; i += 512) { s_w1[i] = a.w1[i]; s_w2[i] = a.w2[i]; }
    }
    __syncthreads();
    // fresh scalar copies of C / Cr for the ticket loop: expressions like C * 4 are otherwise shared with the (divergent) weight-copy loop
    // above, become "uniform values defined under divergent control flow" = VGPRs, and get spilled
    int C_ = a.C, Cr_ = a.Cr;
    asm volatile("" : "+s"(C_), "+s"(Cr_));
    const u32 EP = __builtin_amdgcn_readfirstlane(s_ep);
    const u32 GRAN_TAG = (EP + 1u) ? EP + 1u : 1u;                    // 0 is what a zeroed granule holds
    int par = 0;
    for (;;) {
        __syncthreads();
        // Everything a thread derives from its id is re-derived PER SLICE from an opaque copy: left alone, hipcc hoists each
        // slice-invariant (lane offsets, LDS addresses, predicates ...) out of the ticket loop, and with the row resident in 52 of the
        // 80 registers that three workgroups per CU allow it then spills those invariants to scratch (48 bytes per lane in round 4's
        // <13, ., ., 3> instantiation -- the C2 bench shape).  A handful of integer instructions per slice instead.
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int lane = t & 63, wave = t >> 6;
        const u32 tk = __builtin_amdgcn_readfirstlane(s_tk[par]);      // wave-uniform by construction: image / slab indices, the granule base
        if (tk >= (u32)a.total) return;                                // and the row descriptors live in scalar registers
        const int b = tk / a.gpi, c0 = (tk - b * a.gpi) * ECW;
        const u32 rw = __builtin_amdgcn_readfirstlane((u32)(b * C_ + c0 + wave));
        const long row = (long)rw * a.HW;
        const rsrc_t rx = make_rsrc(a.x + row, (u32)a.HW * 4u), ry = make_rsrc(a.y + row, (u32)a.HW * 4u);
        const u32 voff = (u32)lane * 16u;                              // lanes beyond the row: zeros in, stores dropped (range check)
        v4f r[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            // chunk j sits j KB behind the lane's first: steps 0 .. 3 ride in the instruction's immediate, every further 4 KB in the VGPR
            // offset (formed behind an opaque copy: as scalar offsets they cost nine SGPRs, and past ~100 SGPRs hipcc spills scalars
            // through VGPRs to scratch)
            u32 vo = voff;
            asm volatile("" : "+v"(vo));
            r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + (u32)(j >> 2) * 4096u + (u32)(j & 3) * 1024u, 0, 0));
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { s0 += r[j].x; s1 += r[j].y; s2 += r[j].z; s3 += r[j].w; }
        const float mean = wave_sum_sw((s0 + s1) + (s2 + s3)) * inv;
        u64* gb = a.gran + (long)b * C_;
        if (lane == 0)
            __hip_atomic_store(gb + c0 + wave, ((u64)GRAN_TAG << 32) | (u64)__float_as_uint(mean), AGENT_RLX);

        // sweep the image's granules until every tag is in
        u32 spins = 0;
        bool mine_done = false;                                        // C <= 512: one granule per thread; larger C loops
        const int ts = t;
        for (;;) {
            bool ok = true;
            if (C_ <= 512) {
                if (ts < C_ && !mine_done) {
                    const u64 g = __hip_atomic_load(gb + (u32)ts, AGENT_RLX);     // scalar base + 32-bit lane offset
                    if ((u32)(g >> 32) == GRAN_TAG) { s_p[ts] = __uint_as_float((u32)g); mine_done = true; }
                    else ok = false;
                }
            } else {
                for (int cc = ts; cc < C_; cc += 512) {
                    const u64 g = __hip_atomic_load(gb + cc, AGENT_RLX);
                    if ((u32)(g >> 32) == GRAN_TAG) s_p[cc] = __uint_as_float((u32)g);
                    else ok = false;
                }
            }
            // workgroup-wide AND of `ok` in ONE barrier: a ballot per wave, the eight votes through LDS, two vote rows used alternately (a
            // row is rewritten two sweeps later, when every wave has passed the barrier in between).  __syncthreads_and() costs three
            // barriers per sweep and drags threadIdx.y / .z into registers.
            const int vp = (int)(spins & 1u);
            const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;      // evaluated by the WHOLE wave (not under the lane-0 branch below)
            if ((ts & 63) == 0) s_ok[vp][ts >> 6] = wave_ok ? 1u : 0u;
            __syncthreads();
            const u32 votes = s_ok[vp][0] & s_ok[vp][1] & s_ok[vp][2] & s_ok[vp][3] & s_ok[vp][4] & s_ok[vp][5] & s_ok[vp][6] & s_ok[vp][7];
            if (__builtin_amdgcn_readfirstlane(votes)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > a.spin) {
                if (t == 0) {
                    __hip_atomic_store(a.err, 1u, AGENT_RLX);
                    if (a.herr) __hip_atomic_store(a.herr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
        }
        // Next slice's ticket: only now, when this workgroup no longer waits for anybody -- a workgroup that held an unprocessed
        // ticket of the image it is still waiting for would deadlock.  Its latency hides behind the MLP and the stores.
        if (t == 0) s_tk[par ^ 1] = draw(EP);
        // excitation: h = relu(W1 p) (16 lanes per hidden unit), g = sigmoid(W2[c,:] h) (one wave per channel)
        const float* w1 = WLDS ? s_w1 : a.w1;
        const float* w2 = WLDS ? s_w2 : a.w2;
        const int tm = t;
        const int part = tm & 15, jl = tm >> 4;
        // the LDS bases of h / W1 / W2 as opaque SCALARS formed here (a hoisted VGPR copy of each was what the 80-register build spilled)
        u32 oh = (u32)C_;
        asm volatile("" : "+s"(oh));
        float* s_hl = smem + oh;
        if (WLDS) {
            u32 o1 = (u32)(C_ + Cr_), o2 = (u32)(C_ + Cr_ + Cr_ * C_);
            asm volatile("" : "+s"(o1), "+s"(o2));
            w1 = smem + o1;
            w2 = smem + o2;
        }
        for (int j0 = 0; j0 < Cr_; j0 += 32) {
            const int j = j0 + jl;
            float acc = 0.f;
            if (j < Cr_) {
                const float* wrow = w1 + (long)j * C_;
                for (int cc = part; cc < C_; cc += 16) acc += wrow[cc] * s_p[cc];
            }
            acc += __shfl_xor(acc, 8, WAVE);
            acc += __shfl_xor(acc, 4, WAVE);
            acc += __shfl_xor(acc, 2, WAVE);
            acc += __shfl_xor(acc, 1, WAVE);
            if (part == 0 && j < Cr_) s_hl[j] = relu_nan(EXTRA ? acc + (a.b1 ? a.b1[j] : 0.f) : acc);
        }
        __syncthreads();
        const float* w2r = w2 + (long)(c0 + wave) * Cr_;
        float z = 0.f;
        for (int j = lane; j < Cr_; j += 64) z += w2r[j] * s_hl[j];
        const float g = EXTRA ? se_gate(wave_sum_sw(z) + (a.b2 ? a.b2[c0 + wave] : 0.f), a.gate) : sigmoidf_(wave_sum_sw(z));
        u32 ob = voff;                                                // row step in the VGPR offset of the stores: cbam_single.hip
        asm volatile("" : "+v"(ob));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const v4f o = r[j] * g;
            if (NTS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, AUX_NT);
            else     __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, 0);
        }
        par ^= 1;
    }
}

// launch dispatch over the template parameters; the three-workgroups-per-CU build exists only up to 13 float4 per lane (occ is 2 beyond)
template <int NV, bool EXTRA>
static void se_launch(int occ, bool nts, bool wlds, int grid, size_t smem, hipStream_t st, const SeSingleArgs& a) {
    if constexpr (NV <= 13) {
        if (occ == 3) {
            if (wlds) se_single_kernel<NV, true, true, 3, EXTRA><<<grid, 512, smem, st>>>(a);
            else      se_single_kernel<NV, true, false, 3, EXTRA><<<grid, 512, smem, st>>>(a);
            return;
        }
    }
    if (nts && wlds) se_single_kernel<NV, true, true, 2, EXTRA><<<grid, 512, smem, st>>>(a);
    else if (nts)    se_single_kernel<NV, true, false, 2, EXTRA><<<grid, 512, smem, st>>>(a);
    else if (wlds)   se_single_kernel<NV, false, true, 2, EXTRA><<<grid, 512, smem, st>>>(a);
    else             se_single_kernel<NV, false, false, 2, EXTRA><<<grid, 512, smem, st>>>(a);
}
template <bool EXTRA>
static void se_launch_nv(int nv, int occ, bool nts, bool wlds, int grid, size_t smem, hipStream_t st, const SeSingleArgs& a) {
    if (nv <= 1) se_launch<1, EXTRA>(occ, nts, wlds, grid, smem, st, a);
    else if (nv <= 2) se_launch<2, EXTRA>(occ, nts, wlds, grid, smem, st, a);
    else if (nv <= 4) se_launch<4, EXTRA>(occ, nts, wlds, grid, smem, st, a);
    else if (nv <= 8) se_launch<8, EXTRA>(occ, nts, wlds, grid, smem, st, a);
    else if (nv <= 13) se_launch<13, EXTRA>(occ, nts, wlds, grid, smem, st, a);
    else se_launch<16, EXTRA>(occ, nts, wlds, grid, smem, st, a);
}

}  // namespace

namespace mi355 {

// ticket | err words of the single-read SE kernel (kept at B + 2 words: the layout the bindings read the error word from)
size_t fused_state_bytes(int B) { return (((size_t)B + 2) * sizeof(u32) + 15) & ~(size_t)15; }

bool eca_single_applicable(int C, int k, int H, int W) {
    const long HW = (long)H * W;
    return opt_eca_single() && (HW % 4 == 0) && (HW / 4 <= 16 * 64) && (C % ECW == 0) && (k - 1 <= 8) && (k & 1);
}

int eca_single(const float* x, const float* taps, float* y, int B, int C, int k, int H, int W, hipStream_t st) {
    const int HW = H * W, n4 = HW / 4, gpi = C / ECW;
    const long total_l = (long)B * gpi;
    if (total_l > (1L << 30)) return fail(MI355_EUNSUPPORTED, "eca_single: too many slices");
    const int total = (int)total_l, per_xcd = (total + 7) / 8, grid = per_xcd * 8;
    const int nv = (n4 + 63) / 64;
    const long nt = opt_nt();
    // loads are always plain: the halo rows are the neighbours' resident rows and should stay in the XCD's L2 (measured on MI355X
    // at the C2 shape: 0.307 ms plain vs 0.358 ms with non-temporal loads); "nt" bit1 still selects non-temporal stores.
#define GO(NV_)                                                                                                        \
    do {                                                                                                               \
        if (nt & 2) eca_halo_kernel<NV_, true><<<grid, 512, 0, st>>>(x, taps, y, C, k, HW, gpi, total, per_xcd);   \
        else        eca_halo_kernel<NV_, false><<<grid, 512, 0, st>>>(x, taps, y, C, k, HW, gpi, total, per_xcd);  \
    } while (0)
    if (nv <= 1) GO(1);
    else if (nv <= 2) GO(2);
    else if (nv <= 4) GO(4);
    else if (nv <= 8) GO(8);
    else if (nv <= 13) GO(13);
    else GO(16);
#undef GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MI355_EHIP, "eca_single: launch -> %s", hipGetErrorString(e));
    return MI355_OK;
}

// ---- SE single read (granule exchange) ------------------------------------------------------------------------------------
size_t se_single_extra_bytes(int B, int C) { return (size_t)B * C * sizeof(u64); }

bool se_single_applicable(int C, int Cr, int H, int W) {
    const long HW = (long)H * W;
    // every slice of an image (C / 8 workgroups) has to be resident at the same time, or the image's workgroups wait for granules
    // nobody can publish: at least two workgroups per CU are resident in every configuration of the kernel
    return opt_se_single() && (HW % 4 == 0) && (HW / 4 <= 16 * 64) && (C % ECW == 0) && ((size_t)(C + Cr) * 4 <= 48 * 1024) &&
           C / ECW <= resident_slots(2);
}

// `state` = epoch | (B - 1 unused words) | ticket | err (fused_state_bytes), `gran` = B*C granules (se_single_extra_bytes)
int se_single(const float* x, const float* w1, const float* w2, float* y, int B, int C, int Cr, int H, int W, void* state,
              void* gran, SeExtra ex, hipStream_t st) {
    SeSingleArgs a{};
    a.x = x; a.y = y; a.w1 = w1; a.w2 = w2; a.b1 = ex.b1; a.b2 = ex.b2; a.gate = ex.gate;
    a.gran = static_cast<u64*>(gran);
    a.ticket = static_cast<u32*>(state) + B;
    a.epoch = static_cast<u32*>(state);
    a.err = a.ticket + 1;
    a.herr = sync_err_word_on(st); a.spin = spin_limit();
    if (int rc = sync_pending("se_single")) return rc;
    a.C = C; a.Cr = Cr; a.HW = H * W; a.n4 = a.HW / 4; a.gpi = C / ECW;
    a.inv = 1.0f / (float)a.HW;
    const long total_l = (long)B * a.gpi;
    if (total_l > (1L << 30)) return fail(MI355_EUNSUPPORTED, "se_single: too many slices");
    a.total = (int)total_l;
    const bool wlds = (size_t)2 * C * Cr * sizeof(float) <= 48 * 1024;   // both weight matrices resident in LDS
    const int nv = (a.n4 + 63) / 64;
    // three workgroups per CU (<= 80 VGPRs) only while the row leaves room beside it: 13 float4 per lane (56 x 56) is the limit,
    // 16 (64 x 64) would spill 100+ bytes per lane and runs two per CU at 112 registers instead
    const int occ = (opt_se_occ() == 3 && nv <= 13 && (!wlds || (size_t)(C + Cr + 2 * C * Cr) * 4 <= 50 * 1024)) ? 3 : 2;
    long grid = (long)resident_slots(occ);                // 512-thread workgroups per CU: 2 (<= 128 VGPRs) or 3 (<= 80)
    if (a.gpi > grid) return fail(MI355_EUNSUPPORTED, "se_single: an image needs %d resident workgroups, the device holds %ld", a.gpi, grid);
    if (grid > a.total) grid = a.total;
    const unsigned long long key = ((unsigned long long)B << 32) ^ (unsigned long long)C ^ ((unsigned long long)grid << 44) ^ 0x5E00000000000000ull;
    hipError_t e = hipSuccess;
    if (!ws_known(state, key, st)) {                      // unknown history (first use, other shape, "ws_persistent" off): epoch, ticket, granules
        e = ws_zero_async(state, fused_state_bytes(B) + se_single_extra_bytes(B, C), st);     // state | granules are contiguous (chan_attn.hip)
        if (e != hipSuccess) { ws_forget(state); return fail(MI355_EHIP, "se_single: zeroing -> %s", hipGetErrorString(e)); }
    }
    const size_t smem = (size_t)(C + Cr + (wlds ? 2 * C * Cr : 0)) * sizeof(float);
    const bool nts = (opt_nt() & 2) != 0;
    const bool extra = ex.b1 || ex.b2 || ex.gate;
    if (extra) se_launch_nv<true>(nv, occ, nts, wlds, (int)grid, smem, st, a);
    else       se_launch_nv<false>(nv, occ, nts, wlds, (int)grid, smem, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) { ws_forget(state); return fail(MI355_EHIP, "se_single: launch -> %s", hipGetErrorString(e)); }
    return MI355_OK;
}

}  // namespace mi355
