// chan_stat.hip -- the "per-channel statistic -> tiny transform -> broadcast scale" members of the reference's attention zoo
//  on the single-read scaffolding of the SE kernel (chan_fused.hip): NCHW fp32, x read once, y written once.
//
//   SIMAM   simam.py:32-41                  y = x * sigmoid(d / (4 (sum_hw d / (HW-1) + lambda)) + 0.5),  d = (x - mean_hw x)^2
//   SRM     srm.py:23-34                    g = sigmoid(BN1d_eval(w[c,0] * mean + w[c,1] * std_unbiased))
//   GCTG    gct.py:23-30                    g = exp(-c/2 * ((m - mean_c m) / sqrt(var_c m + eps))^2),    m = mean_hw x
//   LCT     lct.py:29-39                    g = sigmoid(w * (m - mean_grp m) / sqrt(var_grp m + eps) + b) over the channel's group
//   GCT     gate_channel_module.py:32-50    e = sqrt(sum_hw x^2 + eps) * alpha   (l2)   |   e = sum_hw |x| * alpha   (l1)
//                                           g = 1 + tanh(e * gamma / sqrt(mean_c e^2 + eps) + beta)     (l1: / (mean_c |e| + eps))
//
// A 512-thread workgroup keeps 8 channel rows (one per wave) in registers.  SIMAM and SRM only need their own row.  The others
// need one number per channel of the whole image: it travels as an 8-byte {value, tag} granule exactly like the SE means
// (one write-through store per channel, polled sweeps, ticketed slices, launch tag = the workspace's epoch word + 1: chan_fused.hip).
// Shapes the register layout cannot hold (HW % 4 != 0, HW > 4096, C % 8 != 0) take two plain passes (row statistics, then
// gate + scale); both paths accumulate in fixed orders.
#include "common.h"
#include "bufops.h"

namespace {

using v4f = float __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define AGENT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

enum { M_SIMAM = 1, M_SRM = 2, M_GCTG = 3, M_LCT = 4, M_GCT2 = 5, M_GCT1 = 6 };
constexpr int ECW = 8;

struct StatArgs {
    const float* x; float* y;
    const float* p0; const float* p1; const float* p2; const float* p3; const float* p4;   // per-channel parameter arrays (per mode)
    float f0, f1;                         // lambda | bn eps | eps, c | eps | epsilon
    int i0;                               // LCT: channels per group;  GCT1: after_relu
    u64* gran; u32* ticket; u32* epoch; u32* err; u32* herr; float* stats;   // exchange area (single read) / row statistics (two pass)
    int B, C, HW, n4, gpi, total;
    u32 spin;
};

// gate of channel c from its own statistics and (exchange modes) the image's per-channel values s_p[0..C)
// red0/red1: image-level reductions prepared by the caller (GCTG: mean, var of the channel means; GCT: mean_c e^2 or mean_c |e|)
template <int MODE>
__device__ __forceinline__ float gate_of(const StatArgs& a, int c, float mean, float cvar_sum, float own, float red0, float red1) {
    if (MODE == M_SRM) {
        const float stdv = sqrtf(cvar_sum / (float)(a.HW - 1));
        const float z = a.p0[2 * c] * mean + a.p0[2 * c + 1] * stdv;
        const float bn = (z - a.p3[c]) / sqrtf(a.p4[c] + a.f0) * a.p1[c] + a.p2[c];
        return sigmoidf_(bn);
    }
    if (MODE == M_GCTG) {
        const float yn = (own - red0) / sqrtf(red1 + a.f0);
        return expf(-(yn * yn / 2.0f * a.f1));
    }
    if (MODE == M_LCT) {
        const float yn = (own - red0) / sqrtf(red1 + a.f0);
        return sigmoidf_(a.p0[c] * yn + a.p1[c]);
    }
    if (MODE == M_GCT2) {
        const float e = sqrtf(own + a.f0) * a.p0[c];
        const float norm = a.p1[c] / sqrtf(red0 + a.f0);
        return 1.0f + tanhf(e * norm + a.p2[c]);
    }
    if (MODE == M_GCT1) {
        const float e = own * a.p0[c];
        const float norm = a.p1[c] / (red0 + a.f0);
        return 1.0f + tanhf(e * norm + a.p2[c]);
    }
    return 1.0f;
}

// image-level reductions over the C published values in s_p, by the whole workgroup (NT threads), fixed order:
//   GCTG: red0 = mean_c, red1 = mean_c(v^2) - mean_c^2;   GCT2: red0 = mean_c((v + eps) alpha^2);   GCT1: red0 = mean_c |v alpha|
template <int MODE, int NT>
__device__ __forceinline__ void image_reduce(const StatArgs& a, const float* s_p, float* s_red, float& red0, float& red1) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));                                       // formed here, not hoisted out of the caller's slice loop
    const int lane = t & 63, wave = t >> 6;
    float u = 0.f, w = 0.f;
    for (int c = t; c < a.C; c += NT) {
        const float v = s_p[c];
        if (MODE == M_GCTG) { u += v; w += v * v; }
        if (MODE == M_GCT2) { const float e = sqrtf(v + a.f0) * a.p0[c]; u += e * e; }
        if (MODE == M_GCT1) { u += fabsf(v * a.p0[c]); }
    }
    u = wave_sum_sw(u);
    w = wave_sum_sw(w);
    if (lane == 0) { s_red[wave] = u; s_red[16 + wave] = w; }
    __syncthreads();
    float su = 0.f, sw = 0.f;
    for (int i = 0; i < NT / 64; ++i) { su += s_red[i]; sw += s_red[16 + i]; }
    __syncthreads();
    red0 = su / (float)a.C;
    red1 = (MODE == M_GCTG) ? sw / (float)a.C - red0 * red0 : 0.f;
}

// LCT: mean / variance of the published means over the group of channel c (cpg channels), by one wave
__device__ __forceinline__ void group_reduce(const float* s_p, int c, int cpg, float& red0, float& red1) {
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, g0 = (c / cpg) * cpg;
    float u = 0.f, w = 0.f;
    for (int i = lane; i < cpg; i += 64) { const float v = s_p[g0 + i]; u += v; w += v * v; }
    u = wave_sum_sw(u);
    w = wave_sum_sw(w);
    red0 = u / (float)cpg;
    red1 = w / (float)cpg - red0 * red0;
}

template <int MODE, int NV, bool NTS>
__global__ __launch_bounds__(512, (MODE == M_SIMAM ? (NV > 13 ? 2 : 4) : (NV > 13 ? 4 : 6))) void stat_single_kernel(const StatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];       // exchange modes: the image's C published values
    __shared__ float s_red[32];
    __shared__ u32 s_tk[2];
    __shared__ u32 s_ok[2][8];                                        // per-wave votes of the granule sweep, double-buffered by sweep parity
    const int t0 = threadIdx.x;
    constexpr bool XCH = MODE >= M_GCTG;
    // Launch state lives in the workspace, exactly as in se_single_kernel (chan_fused.hip): the tag of a launch is `epoch + 1`, a launch
    // draws total + gridDim.x tickets (one per slice, one stop ticket per workgroup) and the workgroup that draws the last one sets the
    // ticket word back to zero and advances the epoch -- nothing about a launch is a kernel argument, so a hipGraph replay is just
    // another launch and eager launches and replays can share a workspace.
    __shared__ u32 s_ep;
    const u32 last_draw = (u32)a.total + gridDim.x - 1u;
    auto draw = [&](u32 ep) -> u32 {
        const u32 v = __hip_atomic_fetch_add(a.ticket, 1u, AGENT_RLX);
        if (v == last_draw) {
            u32 e1;                                                    // formed here: a hoisted VGPR copy of epoch + 1 would live for the whole kernel
            asm volatile("v_mov_b32 %0, %1" : "=v"(e1) : "s"(ep + 1u));
            __hip_atomic_store(a.ticket, 0u, AGENT_RLX);
            __hip_atomic_store(a.epoch, e1, AGENT_RLX);
        }
        return v;
    };
    u32 EP = 0u, TAG = 1u;
    if (XCH) {
        if (t0 == 0) {
            // acquire: the draw below must not be performed before this load (the epoch cannot move until this workgroup has drawn
            // its stop ticket, but only if the load really comes first)
            const u32 ep = __hip_atomic_load(a.epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            s_ep = ep;
            s_tk[0] = draw(ep);
        }
        __syncthreads();
        EP = __builtin_amdgcn_readfirstlane(s_ep);
        TAG = (EP + 1u) ? EP + 1u : 1u;                               // 0 is what a zeroed granule holds
    }
    int par = 0;
    u32 slice = blockIdx.x;
    for (;;) {
        // thread-id arithmetic is re-derived per slice from an opaque copy (see se_single_kernel, chan_fused.hip: hoisted slice invariants
        // are what the 80-register builds of this family spilled); the ticket is wave-uniform by construction
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int lane = t & 63, wave = t >> 6;
        u32 tk;
        if (XCH) {
            __syncthreads();
            tk = __builtin_amdgcn_readfirstlane(s_tk[par]);
        } else {
            tk = slice;                                               // no waiting between workgroups: a plain grid-stride walk
            slice += gridDim.x;
        }
        if (tk >= (u32)a.total) return;
        const int b = tk / a.gpi, c0 = (tk - b * a.gpi) * ECW, c = c0 + wave;
        // row of this wave through a buffer descriptor: lanes beyond the row read zeros and their stores are dropped (range check),
        // so neither the loads nor the stores carry predicates
        const u32 rlo = __builtin_amdgcn_readfirstlane((u32)((b * a.C + c) & 0x7FFFFFFF));
        const long row = (long)rlo * a.HW;
        const rsrc_t rx = make_rsrc(a.x + row, (u32)a.HW * 4u), ry = make_rsrc(a.y + row, (u32)a.HW * 4u);
        const u32 voff = (u32)lane * 16u;
        v4f r[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (u32)j * 1024u, 0));
        // ---- row statistics from registers --------------------------------------------------------------------------------
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (MODE == M_GCT2) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s0 += r[j].x * r[j].x; s1 += r[j].y * r[j].y; s2 += r[j].z * r[j].z; s3 += r[j].w * r[j].w; }
        } else if (MODE == M_GCT1 && !a.i0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s0 += fabsf(r[j].x); s1 += fabsf(r[j].y); s2 += fabsf(r[j].z); s3 += fabsf(r[j].w); }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s0 += r[j].x; s1 += r[j].y; s2 += r[j].z; s3 += r[j].w; }
        }
        const float tot = wave_sum_sw((s0 + s1) + (s2 + s3));
        const float mean = tot / (float)a.HW;
        float cvs = 0.f;                                              // sum_hw (x - mean)^2  (SIMAM, SRM)
        if (MODE == M_SIMAM || MODE == M_SRM) {
            // lanes beyond the row hold zeros: park them on the mean so that they add nothing (they are never stored)
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (64 * (j + 1) > a.n4) {                            // wave-uniform: only the last occupied slot(s) can be ragged
                    const bool in = lane + 64 * j < a.n4;
                    r[j].x = in ? r[j].x : mean; r[j].y = in ? r[j].y : mean; r[j].z = in ? r[j].z : mean; r[j].w = in ? r[j].w : mean;
                }
            float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const v4f d = r[j] - mean;
                q0 += d.x * d.x; q1 += d.y * d.y; q2 += d.z * d.z; q3 += d.w * d.w;
            }
            cvs = wave_sum_sw((q0 + q1) + (q2 + q3));
        }
        const float own = (MODE == M_GCT2 || MODE == M_GCT1) ? tot : mean;
        float red0 = 0.f, red1 = 0.f;
        if (XCH) {
            u64* gb = a.gran + (long)b * a.C;
            if (lane == 0) __hip_atomic_store(gb + c, ((u64)TAG << 32) | (u64)__float_as_uint(own), AGENT_RLX);
            u32 spins = 0;
            bool timeout = false;
            for (;;) {
                bool ok = true;
                for (int cc = t; cc < a.C; cc += 512) {
                    const u64 g = __hip_atomic_load(gb + cc, AGENT_RLX);
                    if ((u32)(g >> 32) == TAG) s_p[cc] = __uint_as_float((u32)g);
                    else ok = false;
                }
                const int vp = (int)(spins & 1u);                     // one-barrier AND of `ok` (chan_fused.hip se_single_kernel)
                const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
                if (lane == 0) s_ok[vp][wave] = wave_ok ? 1u : 0u;
                __syncthreads();
                const u32 votes = s_ok[vp][0] & s_ok[vp][1] & s_ok[vp][2] & s_ok[vp][3] & s_ok[vp][4] & s_ok[vp][5] & s_ok[vp][6] & s_ok[vp][7];
                if (__builtin_amdgcn_readfirstlane(votes)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > a.spin) { timeout = true; break; }
            }
            if (t == 0) {                                             // nobody is waited for any more: next ticket
                s_tk[par ^ 1] = draw(EP);
                if (timeout) {
                    __hip_atomic_store(a.err, 1u, AGENT_RLX);
                    if (a.herr) __hip_atomic_store(a.herr, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (MODE == M_LCT) group_reduce(s_p, c, a.i0, red0, red1);
            else image_reduce<MODE, 512>(a, s_p, s_red, red0, red1);
        }
        // ---- scale from registers (the row step rides in the VGPR offset of the stores: see cbam_single.hip on the soffset hazard) ----
        u32 ob = voff;
        asm volatile("" : "+v"(ob));
        if (MODE == M_SIMAM) {
            // per-element gate on the transcendental units: sigmoid(z) = rcp(1 + 2^(-z log2 e)), z = d^2 * (1/den) + 0.5 (one exact
            // division per row; v_exp_f32 / v_rcp_f32 are 1-ulp approximations, far inside the 1e-5 budget of the family)
            const float idn = -1.44269504088896340736f / (4.0f * (cvs / (float)(a.HW - 1) + a.f0));
            const float hb = -0.5f * 1.44269504088896340736f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const v4f d = r[j] - mean;
                v4f o;
                o.x = r[j].x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d.x * d.x * idn + hb));
                o.y = r[j].y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d.y * d.y * idn + hb));
                o.z = r[j].z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d.z * d.z * idn + hb));
                o.w = r[j].w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d.w * d.w * idn + hb));
                if (NTS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, AUX_NT);
                else     __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, 0);
            }
        } else {
            const float g = gate_of<MODE>(a, c, mean, cvs, own, red0, red1);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const v4f o = r[j] * g;
                if (NTS) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, AUX_NT);
                else     __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ob + (u32)j * 1024u, 0, 0);
            }
        }
        par ^= 1;
    }
}

// ---- general two-pass path --------------------------------------------------------------------------------------------------------
// pass 1: stats[row] = {sum, sum (x-mean)^2, sum x^2, sum |x|} for every (image, channel) row, one wave per row
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, long rows, int HW) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * HW;
    float s = 0.f, q = 0.f, ab = 0.f;
    for (int i = lane; i < HW; i += 64) { const float v = p[i]; s += v; q += v * v; ab += fabsf(v); }
    s = wave_sum(s); q = wave_sum(q); ab = wave_sum(ab);
    const float mean = s / (float)HW;
    float cv = 0.f;
    for (int i = lane; i < HW; i += 64) { const float d = p[i] - mean; cv += d * d; }
    cv = wave_sum(cv);
    if (lane == 0) { stats[row * 4] = s; stats[row * 4 + 1] = cv; stats[row * 4 + 2] = q; stats[row * 4 + 3] = ab; }
}

// pass 2: one 256-thread workgroup per (image, 4 channels); exchange modes first rebuild the image's published values in LDS
template <int MODE>
__global__ __launch_bounds__(256) void stat_apply_kernel(const StatArgs a, int groups) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];
    __shared__ float s_red[32];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 4 + wave;
    const float* st = a.stats + (long)b * a.C * 4;
    float red0 = 0.f, red1 = 0.f;
    if (MODE >= M_GCTG) {
        for (int cc = t; cc < a.C; cc += 256) {
            const float sum = st[cc * 4], sq = st[cc * 4 + 2], ab = st[cc * 4 + 3];
            s_p[cc] = (MODE == M_GCT2) ? sq : (MODE == M_GCT1 ? (a.i0 ? sum : ab) : sum / (float)a.HW);
        }
        __syncthreads();
        if (MODE != M_LCT) image_reduce<MODE, 256>(a, s_p, s_red, red0, red1);
    }
    if (c >= a.C) return;
    if (MODE == M_LCT) group_reduce(s_p, c, a.i0, red0, red1);
    const float sum = st[c * 4], cvs = st[c * 4 + 1];
    const float mean = sum / (float)a.HW;
    const float own = (MODE >= M_GCTG) ? s_p[c] : mean;
    const long row = ((long)b * a.C + c) * a.HW;
    if (MODE == M_SIMAM) {
        const float den = 4.0f * (cvs / (float)(a.HW - 1) + a.f0);
        for (int i = lane; i < a.HW; i += 64) {
            const float v = a.x[row + i], d = v - mean;
            a.y[row + i] = v * sigmoidf_(d * d / den + 0.5f);
        }
    } else {
        const float g = gate_of<MODE>(a, c, mean, cvs, own, red0, red1);
        for (int i = lane; i < a.HW; i += 64) a.y[row + i] = a.x[row + i] * g;
    }
}

template <int MODE>
int run(StatArgs a, int H, int W, void* ws, size_t ws_bytes, hipStream_t st) {
    const int B = a.B, C = a.C;
    a.HW = H * W;
    constexpr bool XCH = MODE >= M_GCTG;
    const bool single = mi355::opt_zoo_single() && (a.HW % 4 == 0) && (a.HW / 4 <= 16 * 64) && (C % ECW == 0) && aligned16(a.x) &&
                        aligned16(a.y) && (size_t)C * 4 <= 48 * 1024 && (MODE != M_LCT || a.i0 <= C) &&
                        (!XCH || C / ECW <= mi355::resident_slots(2));     // an image's slices must all be resident to exchange granules
    if (single) {
        a.n4 = a.HW / 4; a.gpi = C / ECW;
        const long total_l = (long)B * a.gpi;
        if (total_l > (1L << 30)) return mi355::fail(MI355_EUNSUPPORTED, "channel-statistics gate: too many slices");
        a.total = (int)total_l;
        const int nv = (a.n4 + 63) / 64;
        // workgroups per CU = what the kernel's launch bounds allow: three (<= 80 VGPRs) up to 13 float4 per lane, two beyond; SimAM's
        // per-element gate needs 120 registers (two per CU), 141 at 16 float4 per lane (one per CU)
        long grid = (long)mi355::resident_slots(MODE == M_SIMAM ? (nv > 13 ? 1 : 2) : (nv > 13 ? 2 : 3));
        if (grid > a.total) grid = a.total;
        if (XCH) {
            if (ws_bytes < mi355::zoo_workspace_bytes(B, C)) return mi355::fail(MI355_EINVAL, "channel-statistics gate: workspace too small");
            char* base = static_cast<char*>(ws);
            a.ticket = reinterpret_cast<u32*>(base);
            a.err = a.ticket + 1;
            a.epoch = a.ticket + 2;
            a.herr = mi355::sync_err_word_on(st); a.spin = mi355::spin_limit();
            if (int rc = mi355::sync_pending("channel-statistics gate")) return rc;
            a.gran = reinterpret_cast<u64*>(base + 16);
            // the grid size is part of the key: the ticket protocol counts total + grid draws per launch
            const unsigned long long key = ((unsigned long long)B << 32) ^ (unsigned long long)C ^ ((unsigned long long)grid << 44) ^ ((unsigned long long)MODE << 56);
            if (!mi355::ws_known(ws, key, st)) {          // unknown history: ticket, epoch, granules (a kernel, not a memset node: api.hip ws_zero_async)
                hipError_t e = mi355::ws_zero_async(ws, 16 + (size_t)B * C * 8, st);
                if (e != hipSuccess) { mi355::ws_forget(ws); return mi355::fail(MI355_EHIP, "channel-statistics gate: zeroing -> %s", hipGetErrorString(e)); }
            }
        }
        const size_t smem = XCH ? (size_t)C * 4 : 0;
        const bool nts = (mi355::opt_nt() & 2) != 0;
#define GO(NV_)                                                                                        \
        do {                                                                                           \
            if (nts) stat_single_kernel<MODE, NV_, true><<<(int)grid, 512, smem, st>>>(a);             \
            else     stat_single_kernel<MODE, NV_, false><<<(int)grid, 512, smem, st>>>(a);            \
        } while (0)
        if (nv <= 2) GO(2);
        else if (nv <= 4) GO(4);
        else if (nv <= 8) GO(8);
        else if (nv <= 13) GO(13);
        else GO(16);
#undef GO
    } else {
        if (ws_bytes < mi355::zoo_workspace_bytes(B, C)) return mi355::fail(MI355_EINVAL, "channel-statistics gate: workspace too small");
        if ((size_t)C * 4 > 64 * 1024) return mi355::fail(MI355_EUNSUPPORTED, "channel-statistics gate: C = %d too large", C);
        a.stats = reinterpret_cast<float*>(static_cast<char*>(ws) + 16 + (size_t)B * C * 8);
        if (XCH) mi355::ws_forget(ws);
        const long rows = (long)B * C;
        row_stats_kernel<<<(int)((rows + 3) / 4), 256, 0, st>>>(a.x, a.stats, rows, a.HW);
        const int groups = (C + 3) / 4;
        stat_apply_kernel<MODE><<<B * groups, 256, XCH ? (size_t)C * 4 : 0, st>>>(a, groups);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (XCH) mi355::ws_forget(ws); return mi355::fail(MI355_EHIP, "channel-statistics gate: launch -> %s", hipGetErrorString(e)); }
    return MI355_OK;
}

}  // namespace

namespace mi355 {
// ticket, err, epoch, pad (16 B) | granules B*C*8 | row statistics of the two-pass path B*C*16
size_t zoo_workspace_bytes(int B, int C) { return 16 + (size_t)B * C * 8 + (size_t)B * C * 16; }
}  // namespace mi355

extern "C" {

size_t mi355_chan_stat_workspace_bytes(int B, int C) { return mi355::zoo_workspace_bytes(B, C); }

int mi355_simam_fwd(const float* x, float* y, int B, int C, int H, int W, float e_lambda, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws && B > 0 && C > 0 && H > 0 && W > 0 && (long)H * W > 1);
    StatArgs a{};
    a.x = x; a.y = y; a.B = B; a.C = C; a.f0 = e_lambda;
    return run<M_SIMAM>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_srm_fwd(const float* x, const float* cfc, const float* bn_weight, const float* bn_bias, const float* bn_mean, const float* bn_var,
                  float bn_eps, float* y, int B, int C, int H, int W, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws && cfc && bn_weight && bn_bias && bn_mean && bn_var && B > 0 && C > 0 && H > 0 && W > 0 && (long)H * W > 1);
    StatArgs a{};
    a.x = x; a.y = y; a.B = B; a.C = C; a.p0 = cfc; a.p1 = bn_weight; a.p2 = bn_bias; a.p3 = bn_mean; a.p4 = bn_var; a.f0 = bn_eps;
    return run<M_SRM>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_gct_gauss_fwd(const float* x, float* y, int B, int C, int H, int W, float c, float eps, void* ws, size_t ws_bytes,
                        mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws && B > 0 && C > 0 && H > 0 && W > 0);
    StatArgs a{};
    a.x = x; a.y = y; a.B = B; a.C = C; a.f0 = eps; a.f1 = c;
    return run<M_GCTG>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_lct_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int groups, int H, int W, float eps, void* ws,
                  size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws && w && b && B > 0 && C > 0 && groups > 0 && C % groups == 0 && H > 0 && W > 0);
    StatArgs a{};
    a.x = x; a.y = y; a.B = B; a.C = C; a.p0 = w; a.p1 = b; a.f0 = eps; a.i0 = C / groups;
    return run<M_LCT>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_gct_fwd(const float* x, const float* alpha, const float* gamma, const float* beta, float* y, int B, int C, int H, int W,
                  float epsilon, int mode_l1, int after_relu, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws && alpha && gamma && beta && B > 0 && C > 0 && H > 0 && W > 0);
    StatArgs a{};
    a.x = x; a.y = y; a.B = B; a.C = C; a.p0 = alpha; a.p1 = gamma; a.p2 = beta; a.f0 = epsilon; a.i0 = after_relu ? 1 : 0;
    return mode_l1 ? run<M_GCT1>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream))
                   : run<M_GCT2>(a, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

}  // extern "C"
