// cbam_single.hip -- CBAM (channel gate, then spatial gate) with x read from HBM ONCE and y written once (gfx950).
//
// The multi-pass path (chan_attn.hip) streams x three times: pool, per-pixel statistics, apply.  Here a 512-thread workgroup
// owns a BAND of R image rows for ALL channels (R*W <= 128 pixels; 2 x 56 at the C2 shape = 114 KB) and keeps it in
// registers from the first load to the final store.  Thread layout: SEG lanes (16 or 32) per channel segment -- lane q holds
// pixel quad q of the band, the 512/SEG segment groups take channels round-robin -- so a channel's band is one contiguous,
// coalesced run and every per-channel reduction is a butterfly inside one segment.
//
// Three things an image's bands must tell each other travel as self-validating 16-byte granules {v0, tag, v1, tag}, each
// written by ONE write-through (sc1) store and polled with sc1 loads (MI355X_MICROARCH.md price list: handoff / allgather):
//   hop 1  per-channel (sum, max) of every band          g1[b][band][c]   -> band k adds up channels k*cpb .. (k+1)*cpb-1
//   hop 2  per-channel (avg, max) of the image           g2[b][c]         -> every band: excitation MLP -> gc[c]
//   hop 3  per-pixel (mean_c, max_c) of x*gc per row     g3[b][pixel]     -> neighbours' halo rows of the k x k gate conv
// Every sum is accumulated in a fixed order (butterfly inside the segment, bands in band order, channel groups in group
// order), so results are run-to-run identical and independent of placement.  Slices (image, band) are handed out in image
// order by a ticket; a workgroup only takes its next ticket once it no longer waits for anybody, so progress needs only
// NB (bands per image) running workgroups, never a particular placement.  The tag of a launch is `epoch + 1`, read from the
// workspace; the workgroup that draws the last ticket of a launch (total + gridDim.x draws: one per slice, one stop ticket per
// workgroup) zeroes the ticket word and advances the epoch.  Slots written by earlier launches never look valid, the region is
// only zeroed when its history is unknown, and because nothing about a launch lives on the host, eager launches and hipGraph
// replays can share a workspace.  Polls are bounded and raise the workspace error word instead of hanging.
#include "common.h"
#include "bufops.h"

namespace {

using v4f = float __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define AGENT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void gran_put(rsrc_t g, u32 idx, float v0, float v1, u32 TAG) {
    const u32x4 v = {__float_as_uint(v0), TAG, __float_as_uint(v1), TAG};
    __builtin_amdgcn_raw_buffer_store_b128(v, g, idx * 16u, 0, AUX_SC1);      // one write-through 16-byte store
}
__device__ __forceinline__ bool gran_get(rsrc_t g, u32 idx, float& v0, float& v1, u32 TAG) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(g, idx * 16u, 0, AUX_SC1);
    v0 = __uint_as_float(v.x);
    v1 = __uint_as_float(v.z);
    return v.y == TAG && v.w == TAG;
}

// Cross-lane steps as DPP modifiers (one VALU op each, no LDS round trip like ds_bpermute): quad_perm for xor 1 / xor 2,
// row_ror for the rotations inside a row of 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    // old = the source itself and bound_ctrl set: every lane of these permutations has a valid source, so no "old" value has to be
    // materialised and the compiler can fold the permutation into the consuming add / max as a DPP modifier
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rdlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Sum / max over the SEG (16 or 32) lanes of a segment; the result is only guaranteed in the segment's lane 0 (fixed order).
template <int SEG>
__device__ __forceinline__ void seg_reduce(float& s, float& m, int lane) {
    s += dpp<0xB1>(s);  m = fmaxf(m, dpp<0xB1>(m));                   // xor 1
    s += dpp<0x4E>(s);  m = fmaxf(m, dpp<0x4E>(m));                   // xor 2
    s += dpp<0x124>(s); m = fmaxf(m, dpp<0x124>(m));                  // row_ror 4
    s += dpp<0x128>(s); m = fmaxf(m, dpp<0x128>(m));                  // row_ror 8: every lane of the row holds the row total
    if (SEG == 32) {                                                  // rows 0+1 -> lane 0, rows 2+3 -> lane 32 (only those two lanes are used)
        const float s1 = rdlane(s, 16), s3 = rdlane(s, 48), m1 = rdlane(m, 16), m3 = rdlane(m, 48);
        s += (lane < 32) ? s1 : s3;
        m = fmaxf(m, (lane < 32) ? m1 : m3);
    }
}

struct CbamSingleArgs {
    const float* x; float* y; const float* w1; const float* w2; const float* wconv;
    u32x4* g1; u32x4* g2; u32x4* g3; u32* ticket; u32* epoch; u32* err; u32* herr;   // herr: pinned host word every later call checks (api.hip)
    u32 spin;
    int C, Cr, H, W, ks, R, Q, NB, cpb, total, nts, wlds;
#ifdef CBAM_TIMING
    unsigned long long* dbg;                                          // [gridDim][16 slices][10 stamps] of wall_clock64 (tools/cbam_timing.hip)
#endif
};

#ifdef CBAM_TIMING
#define STAMP(k)                                                                                                  \
    do {                                                                                                          \
        if (t == 0 && a.dbg && nslice < 16) a.dbg[((long)blockIdx.x * 16 + nslice) * 10 + (k)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <int NT, int SEG, int NV, bool FULL>
__global__ __launch_bounds__(NT, 4) void cbam_single_kernel(const CbamSingleArgs a) {
    constexpr int CL = NT / SEG;                                     // channel groups (segments) per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ u32 s_tk[2];
    __shared__ u32 s_ep;
    const int C = a.C, Cr = a.Cr, W = a.W, H = a.H, ks = a.ks, pad = (ks - 1) >> 1;
    const int HW = H * W, npx = a.R * W, TW = W + 2 * pad, TH = a.R + 2 * pad;
    const int Cp = (CL * NV > C ? CL * NV : ((C + 3) & ~3)), Crp = (Cr + 3) & ~3, L2p = (a.NB * a.cpb + 3) & ~3;   // every sub-array starts 16-byte aligned
    float* s_a = smem;                                                // avg[C]
    float* s_m = s_a + Cp;                                            // max[C]
    float* s_gc = s_m + Cp;                                           // channel gates [C]
    float* s_h = s_gc + Cp;                                           // hidden: relu(W1 avg)[Crp] | relu(W1 max)[Crp]
    float* s_l2s = s_h + 2 * Crp;                                     // hop-1 landing: sums [NB*cpb]
    float* s_l2m = s_l2s + L2p;                                       //                maxima
    float* s_ps = s_l2m + L2p;                                        // per-group pixel partial sums [CL][SEG*4]
    float* s_pm = s_ps + CL * SEG * 4;                                // per-group pixel partial maxima
    float* s_t = s_pm + CL * SEG * 4;                                 // statistics tile [2][TH][TW], zero border
    float* s_gs = s_t + ((2 * TH * TW + 3) & ~3);                     // spatial gate of the band [SEG*4 >= npx]
    float* s_wc = s_gs + SEG * 4;                                     // conv taps [2*ks*ks (pad 4)]
    float* s_w1 = s_wc + ((2 * ks * ks + 3) & ~3);                    // (wlds) W1 [Cr*C] | W2 [C*Cr]
    float* s_w2 = s_w1 + Cr * C;

    const int t = threadIdx.x, q = t & (SEG - 1), cl = t / SEG;
    const bool qa = q < a.Q;
    // one ticket; the last draw of the launch (number total + gridDim.x - 1) resets the ticket word and advances the epoch
    const u32 last_draw = (u32)a.total + gridDim.x - 1u;
    auto draw = [&](u32 ep) -> u32 {
        const u32 v = __hip_atomic_fetch_add(a.ticket, 1u, AGENT_RLX);
        if (v == last_draw) {
            __hip_atomic_store(a.ticket, 0u, AGENT_RLX);
            __hip_atomic_store(a.epoch, ep + 1u, AGENT_RLX);
        }
        return v;
    };
    if (t == 0) {
        // before the first draw (acquire: the fetch_add below may not be performed ahead of this load): the epoch cannot move until
        // this workgroup has drawn its stop ticket
        const u32 ep = __hip_atomic_load(a.epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        s_ep = ep;
        s_tk[0] = draw(ep);
    }
    for (int i = t; i < 2 * ks * ks; i += NT) s_wc[i] = a.wconv[i];
    if (a.wlds)
        for (int i = t; i < Cr * C; i += NT) { s_w1[i] = a.w1[i]; s_w2[i] = a.w2[i]; }
    const float* w1 = a.wlds ? s_w1 : a.w1;
    const float* w2 = a.wlds ? s_w2 : a.w2;
    __syncthreads();
    const u32 EP = s_ep;
    const u32 TAG = (EP + 1u) ? EP + 1u : 1u;                         // 0 is what a zeroed granule holds
    int par = 0;
#ifdef CBAM_TIMING
    int nslice = -1;
#endif

    for (;;) {
        __syncthreads();
        const u32 tk = s_tk[par];
        if (tk >= (u32)a.total) return;
#ifdef CBAM_TIMING
        ++nslice;
#endif
        STAMP(0);
        const int b = tk / a.NB, band = tk - b * a.NB, r0 = band * a.R;
        const long img = (long)b * C * HW + (long)r0 * W;
        u32 spins = 0;
        bool timeout = false;

        // ---- the band of every channel -> registers ---------------------------------------------------------------------
        v4f r[NV];
        const u32 ext = ((u32)C * (u32)HW - (u32)r0 * (u32)W) * 4u;                  // bytes from the band start to the image end
        const rsrc_t rx = make_rsrc(a.x + img, ext), ry = make_rsrc(a.y + img, ext);
        const u32 off0 = qa ? ((u32)cl * (u32)HW + 4u * (u32)q) * 4u : OOB, offs = (u32)CL * (u32)HW * 4u;
        // FULL (C == CL*NV): the channel step rides in the SGPR offset.  Otherwise channels >= C are sent out of range per lane
        // (the range check looks at the VGPR offset only, never at the SGPR offset).
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (FULL) r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, off0, (u32)j * offs, 0));
            else r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, (cl + CL * j < C) ? off0 + (u32)j * offs : OOB, 0, 0));
        }
        const rsrc_t rg1 = make_rsrc(a.g1 + (long)b * a.NB * C, (u32)a.NB * (u32)C * 16u);
        const rsrc_t rg2 = make_rsrc(a.g2 + (long)b * C, (u32)C * 16u);
        const rsrc_t rg3 = make_rsrc(a.g3 + (long)b * HW, (u32)HW * 16u);
        // zero the statistics tile while the loads fly
        for (int i = t; i < 2 * TH * TW; i += NT) s_t[i] = 0.f;

        // ---- hop 1, publish: (sum, max) of this band for every channel ------------------------------------------------------
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = cl + CL * j;
            float s = (r[j].x + r[j].y) + (r[j].z + r[j].w);
            float m = qa ? fmaxf(fmaxf(r[j].x, r[j].y), fmaxf(r[j].z, r[j].w)) : -INFINITY;
            seg_reduce<SEG>(s, m, t & 63);
            if (q == 0 && c < C) gran_put(rg1, (u32)(band * C + c), s, m, TAG);
        }
        STAMP(1);                                                                    // loads landed, band partials published
        // ---- hop 1, consume: this band adds up channels ck0 .. ck0+nch-1 over all bands, publishes (avg, max) as hop 2 ------
        const int ck0 = band * a.cpb;
        const int nch = (ck0 >= C) ? 0 : ((C - ck0 < a.cpb) ? C - ck0 : a.cpb);
        const int n1 = a.NB * nch;
        for (;;) {
            bool ok = true;
            for (int i = t; i < n1; i += NT) {
                const int bb = i / nch, cc = i - bb * nch;
                float v0, v1;
                if (gran_get(rg1, (u32)(bb * C + ck0 + cc), v0, v1, TAG)) { s_l2s[bb * a.cpb + cc] = v0; s_l2m[bb * a.cpb + cc] = v1; }
                else ok = false;
            }
            if (__syncthreads_and(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > a.spin) { timeout = true; break; }
        }
        STAMP(2);                                                                    // hop 1 in
        if (t < nch) {
            float s = 0.f, m = -INFINITY;
            for (int bb = 0; bb < a.NB; ++bb) { s += s_l2s[bb * a.cpb + t]; m = fmaxf(m, s_l2m[bb * a.cpb + t]); }
            gran_put(rg2, (u32)(ck0 + t), s / (float)HW, m, TAG);
        }
        // ---- hop 2, consume: (avg, max) of every channel of the image ---------------------------------------------------------
        for (;;) {
            bool ok = true;
            for (int c = t; c < C; c += NT) {
                float v0, v1;
                if (gran_get(rg2, (u32)c, v0, v1, TAG)) { s_a[c] = v0; s_m[c] = v1; }
                else ok = false;
            }
            if (__syncthreads_and(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > a.spin) { timeout = true; break; }
        }
        STAMP(3);                                                                    // hop 2 in
        // ---- channel gates: gc = sigmoid(W2 (relu(W1 avg) + relu(W1 max)))  (cbam.py:31-35) ---------------------------------
        {
            // first half of the threads: W1 avg, second half: W1 max; 16 lanes per hidden unit, NT/32 units per round
            const int half = t / (NT / 2), tt = t & (NT / 2 - 1), part = tt & 15, jl = tt >> 4;
            const float* vec = half ? s_m : s_a;
            float* s_hh = s_h + half * Crp;                           // relu(W1 avg) | relu(W1 max)
            for (int j0 = 0; j0 < Cr; j0 += NT / 32) {
                const int j = j0 + jl;
                float h0 = 0.f, h1 = 0.f;
                if (j < Cr) {
                    const float* wrow = w1 + (long)j * C;
                    int cc = part;
                    for (; cc + 16 < C; cc += 32) { h0 += wrow[cc] * vec[cc]; h1 += wrow[cc + 16] * vec[cc + 16]; }
                    if (cc < C) h0 += wrow[cc] * vec[cc];
                }
                float h = h0 + h1;
                h += dpp<0xB1>(h); h += dpp<0x4E>(h); h += dpp<0x124>(h); h += dpp<0x128>(h);
                if (part == 0 && j < Cr) s_hh[j] = relu_nan(h);
            }
            __syncthreads();
            for (int c = t; c < C; c += NT) {
                const float* w2r = w2 + (long)c * Cr;
                float z0 = 0.f, z1 = 0.f;
                int j = 0;
                for (; j + 1 < Cr; j += 2) {
                    z0 += w2r[j] * (s_h[j] + s_h[Crp + j]);
                    z1 += w2r[j + 1] * (s_h[j + 1] + s_h[Crp + j + 1]);
                }
                if (j < Cr) z0 += w2r[j] * (s_h[j] + s_h[Crp + j]);
                s_gc[c] = sigmoidf_(z0 + z1);
            }
            __syncthreads();
        }
        STAMP(4);                                                                    // gates done
        // ---- per-pixel statistics of x' = x * gc over the channels (cbam.py:43-46) -----------------------------------------------
        {
            v4f ps = {0.f, 0.f, 0.f, 0.f}, pm = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = cl + CL * j;
                if (FULL || c < C) {
                    const v4f v = r[j] * s_gc[c];
                    ps += v;
                    pm.x = fmaxf(pm.x, v.x); pm.y = fmaxf(pm.y, v.y); pm.z = fmaxf(pm.z, v.z); pm.w = fmaxf(pm.w, v.w);
                }
            }
            reinterpret_cast<v4f*>(s_ps)[cl * SEG + q] = ps;
            reinterpret_cast<v4f*>(s_pm)[cl * SEG + q] = pm;
        }
        __syncthreads();
        if (t < npx) {
            float s = 0.f, m = -INFINITY;
#pragma unroll 4
            for (int g = 0; g < CL; ++g) { s += s_ps[g * SEG * 4 + t]; m = fmaxf(m, s_pm[g * SEG * 4 + t]); }
            s = s / (float)C;
            const int ty = t / W, tx = t - ty * W;
            s_t[(0 * TH + ty + pad) * TW + tx + pad] = s;
            s_t[(1 * TH + ty + pad) * TW + tx + pad] = m;
            gran_put(rg3, (u32)(r0 * W + t), s, m, TAG);                                  // hop 3, publish
        }
        STAMP(5);                                                                    // statistics published
        // ---- hop 3, consume: halo rows of the neighbouring bands ------------------------------------------------------------------
        {
            const int up = (r0 < pad) ? r0 : pad;                                    // rows available above
            const int dn = (H - (r0 + a.R) < pad) ? H - (r0 + a.R) : pad;            // rows available below
            const int n3 = (up + dn) * W;
            for (;;) {
                bool ok = true;
                for (int i = t; i < n3; i += NT) {
                    const int hr = i / W, tx = i - hr * W;
                    const int gy = (hr < up) ? r0 - up + hr : r0 + a.R + (hr - up);
                    float v0, v1;
                    if (gran_get(rg3, (u32)(gy * W + tx), v0, v1, TAG)) {
                        const int ty = gy - r0 + pad;
                        s_t[(0 * TH + ty) * TW + tx + pad] = v0;
                        s_t[(1 * TH + ty) * TW + tx + pad] = v1;
                    } else ok = false;
                }
                if (__syncthreads_and(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > a.spin) { timeout = true; break; }
            }
        }
        STAMP(6);                                                                    // hop 3 in
        // nobody is waited for any more: take the next ticket (hidden behind the conv and the stores)
        u32 next_tk = 0;
        if (t == 0) {
            next_tk = draw(EP);                                                      // consumed after the stores below
            if (timeout) {
                __hip_atomic_store(a.err, 1u, AGENT_RLX);
                if (a.herr) __hip_atomic_store(a.herr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        // ---- spatial gate of the band: sigmoid(conv_ks x ks([mean, max]))  (2 -> 1, zero pad, cross-correlation) ----------------
        {
            // four lanes per pixel: lane `part` takes the (plane, dy) tap rows part, part+4, ...; quad butterfly adds them up
            const int p = t >> 2, part = t & 3;
            float acc = 0.f;
            if (p < npx) {
                const int ty = p / W, tx = p - ty * W;
                for (int rr = part; rr < 2 * ks; rr += 4) {
                    const int ch = rr / ks, dy = rr - ch * ks;
                    const float* trow = s_t + (ch * TH + ty + dy) * TW + tx;
                    const float* wrow = s_wc + rr * ks;
                    float a0 = 0.f, a1 = 0.f;
                    int dx = 0;
                    for (; dx + 1 < ks; dx += 2) { a0 += wrow[dx] * trow[dx]; a1 += wrow[dx + 1] * trow[dx + 1]; }
                    if (dx < ks) a0 += wrow[dx] * trow[dx];
                    acc += a0 + a1;
                }
            }
            acc += dpp<0xB1>(acc);
            acc += dpp<0x4E>(acc);
            if (p < npx && part == 0) s_gs[p] = sigmoidf_(acc);
        }
        __syncthreads();
        STAMP(7);                                                                    // spatial gate done
        // ---- y = (x * gc) * gs from registers -------------------------------------------------------------------------------------
        {
            // The channel step goes into the VGPR offset here, not into the SGPR offset: with a REGISTER soffset hipcc (ROCm 7.2)
            // does not pad the ">64-bit store data, then VALU write of the same VGPRs" hazard and the next product overwrote the
            // last dwords of a store still being read (observed on gfx950: .w of lanes 12-15 / 28-31 of every row of 16).
            // `ob` is laundered through an empty asm so the per-j offsets are recomputed here instead of being hoisted out of
            // the slice loop into 16 long-lived VGPRs.
            const v4f s4 = reinterpret_cast<const v4f*>(s_gs)[q & (SEG - 1)];       // lanes beyond the band: stores are dropped (OOB)
            u32 ob = off0;
            asm volatile("" : "+v"(ob));
            if (a.nts) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const v4f v = (r[j] * s_gc[cl + CL * j]) * s4;
                    const u32 vo = (FULL || cl + CL * j < C) ? ob + (u32)j * offs : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, vo, 0, AUX_NT);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const v4f v = (r[j] * s_gc[cl + CL * j]) * s4;
                    const u32 vo = (FULL || cl + CL * j < C) ? ob + (u32)j * offs : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, vo, 0, 0);
                }
            }
        }
        if (t == 0) s_tk[par ^ 1] = next_tk;
        STAMP(8);                                                                    // stores issued
        par ^= 1;
    }
}

struct Geo {
    int NT, R, Q, NB, SEG, CL, NV, cpb;
    size_t smem_base, smem_w;
};

// Rows per band: the most pixels per band with R | H, (R*W) % 4 == 0 and R*W <= NT/4 (four conv lanes per pixel); 0 if none.
int band_rows(int H, int W, int nt) {
    int best = 0;
    for (int R = 1; R <= H; ++R) {
        if (H % R || (R * W) % 4 || R * W > nt / 4) continue;
        best = R;
    }
    return best;
}

bool geometry(int nt, int C, int Cr, int H, int W, int ks, Geo& g) {
    if (!(ks & 1) || ks > 15) return false;
    const int best = band_rows(H, W, nt);
    if (!best) return false;
    g.NT = nt;
    g.R = best;
    g.Q = best * W / 4;
    g.NB = H / best;
    g.SEG = g.Q > 16 ? 32 : 16;
    g.CL = nt / g.SEG;
    const int nv = (C + g.CL - 1) / g.CL;
    if (nv > 16) return false;
    g.NV = nv <= 4 ? 4 : (nv <= 8 ? 8 : 16);
    g.cpb = (C + g.NB - 1) / g.NB;
    const int pad = (ks - 1) / 2, TW = W + 2 * pad, TH = best + 2 * pad, npx = best * W;
    auto r4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t Cp = (size_t)g.CL * g.NV > (size_t)C ? (size_t)g.CL * g.NV : r4(C);
    g.smem_base = (3 * Cp + 2 * r4(Cr) + 2 * r4((size_t)g.NB * g.cpb) + 2 * (size_t)g.CL * g.SEG * 4 +
                   r4(2 * (size_t)TH * TW) + (size_t)g.SEG * 4 + r4(2 * (size_t)ks * ks)) * 4;
    (void)npx;
    g.smem_w = 2 * (size_t)C * Cr * 4;
    return g.smem_base <= 40 * 1024;
}

// 512-thread workgroups, two per CU.  (256-thread workgroups with half-size bands -- four exchange chains per CU instead of two -- were
// measured at 0.72 ms against 0.49 ms at the C2 shape: every phase got slower.  DESIGN.md 6.1.)
bool pick_geometry(int C, int Cr, int H, int W, int ks, Geo& g) { return geometry(512, C, Cr, H, W, ks, g); }

}  // namespace

namespace mi355 {
#ifdef CBAM_TIMING
unsigned long long* g_cbam_dbg = nullptr;
#endif

// extra workspace of the single-read CBAM: g1 | g2 | g3 granules | ticket, err, pad (0 when no band geometry exists)
size_t cbam_single_extra_bytes(int B, int C, int H, int W) {
    const int R = band_rows(H, W, 512);
    if (!R) return 0;
    return ((size_t)B * (H / R) * C + (size_t)B * C + (size_t)B * H * W) * 16 + 16;
}

bool cbam_single_applicable(int C, int Cr, int H, int W, int ks) {
    Geo g;
    // all NB bands of an image must be resident together (two workgroups per CU): ADVICE r1, e.g. H = 1024 at R = 1
    return opt_cbam_single() && pick_geometry(C, Cr, H, W, ks, g) && g.NB <= resident_slots(2);
}

int cbam_single(const float* x, const float* w1, const float* w2, const float* wconv, float* y, int B, int C, int Cr, int H, int W,
                int ks, void* extra, hipStream_t st) {
    Geo g;
    if (!pick_geometry(C, Cr, H, W, ks, g)) return fail(MI355_EUNSUPPORTED, "cbam_single: unsupported shape");
    CbamSingleArgs a{};
    a.x = x; a.y = y; a.w1 = w1; a.w2 = w2; a.wconv = wconv;
    a.g1 = static_cast<u32x4*>(extra);
    a.g2 = a.g1 + (size_t)B * g.NB * C;
    a.g3 = a.g2 + (size_t)B * C;
    a.ticket = reinterpret_cast<u32*>(a.g3 + (size_t)B * H * W);
    a.err = a.ticket + 1;
    a.epoch = a.ticket + 2;
    a.herr = sync_err_word_on(st); a.spin = spin_limit();
    if (int rc = sync_pending("cbam_single")) return rc;
    a.C = C; a.Cr = Cr; a.H = H; a.W = W; a.ks = ks; a.R = g.R; a.Q = g.Q; a.NB = g.NB; a.cpb = g.cpb;
    const long total_l = (long)B * g.NB;
    if (total_l > (1L << 30)) return fail(MI355_EUNSUPPORTED, "cbam_single: too many slices");
    a.total = (int)total_l;
    a.nts = (opt_nt() & 2) ? 1 : 0;
    const bool full = (C == g.CL * g.NV);
    const int per_cu = 2;
    a.wlds = (g.smem_base + g.smem_w <= (size_t)(120 * 1024) / per_cu) ? 1 : 0;
    const size_t smem = g.smem_base + (a.wlds ? g.smem_w : 0);
    long grid = (long)resident_slots(per_cu);                                   // 16 waves per CU at <= 128 VGPRs, <= 120 KB LDS per CU
#ifdef CBAM_TIMING
    a.dbg = g_cbam_dbg;
#endif
    if (g.NB > grid) return fail(MI355_EUNSUPPORTED, "cbam_single: an image needs %d resident workgroups, the device holds %ld", g.NB, grid);
    if (grid > a.total) grid = a.total;
    const unsigned long long key = ((unsigned long long)B << 48) ^ ((unsigned long long)C << 32) ^ ((unsigned long long)H << 16) ^ (unsigned long long)W ^ ((unsigned long long)g.NT << 40) ^ 0xCBA0000000000000ull;
    if (!ws_known(extra, key, st)) {                                            // unknown history: granules, ticket, epoch
        hipError_t e = ws_zero_async(extra, cbam_single_extra_bytes(B, C, H, W), st);
        if (e != hipSuccess) { ws_forget(extra); return fail(MI355_EHIP, "cbam_single: zeroing -> %s", hipGetErrorString(e)); }
    }
#define GO(SEG_, NV_)                                                                              \
    do {                                                                                           \
        if (full) cbam_single_kernel<512, SEG_, NV_, true><<<(int)grid, 512, smem, st>>>(a);       \
        else      cbam_single_kernel<512, SEG_, NV_, false><<<(int)grid, 512, smem, st>>>(a);      \
    } while (0)
    if (g.SEG == 32) {
        if (g.NV == 4) GO(32, 4);
        else if (g.NV == 8) GO(32, 8);
        else GO(32, 16);
    } else {
        if (g.NV == 4) GO(16, 4);
        else if (g.NV == 8) GO(16, 8);
        else GO(16, 16);
    }
#undef GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ws_forget(extra); return fail(MI355_EHIP, "cbam_single: launch -> %s", hipGetErrorString(e)); }
    return MI355_OK;
}

}  // namespace mi355
