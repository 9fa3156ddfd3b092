// chan_attn.hip -- HBM-bound channel / spatial attention family for gfx950: SELayer, ECALayer, CBAM.
//
// Data layout: x, y are NCHW fp32; one (b,c) "row" is HW contiguous floats.  All kernels stream rows with
// 16-byte-per-lane loads (1 KiB per wave instruction) when HW % 4 == 0, and fall back to 4-byte lanes
// otherwise.  Reductions are wave-level __shfl_xor trees (64 lanes) plus LDS partials across the 4 waves
// of a 256-thread workgroup.  The gate math (2-layer excite / k-tap conv / 7x7 stencil) is tiny and is
// recomputed per workgroup from the pooled vectors so that no extra launch sits between the two streaming
// passes.
//
// Algorithmic HBM traffic per image: read x once + write y once = 2*C*H*W*4 bytes.  The
// gate of image b depends on all of image b (3.2 MB at the C2 shape, larger than LDS), so the data is
// streamed twice (SE/ECA) or three times (CBAM); the launcher therefore walks the batch in chunks of
// `chunk_images` images so that the re-read of a chunk is served by the 256 MiB Infinity Cache instead of
// HBM (DESIGN.md section "channel attention").
#include "common.h"

namespace {

constexpr int RPB = 16;   // rows (channels) per workgroup in the scale kernels

using v4f = float __attribute__((ext_vector_type(4)));   // 16-byte lane vector (global_load/store_dwordx4)

// ---------------------------------------------------------------------------------------------------
// K1: per-row global pooling.  Workgroup = (image, group of RPB channels) -- the SAME block <-> rows map as the
// scale kernels, so a row is pooled and later rescaled by workgroups that sit on the same XCD (block id mod 8).
// One wave per row.  avg[row] = sum/HW (true division, like ATen's mean), mx[row] = max (CBAM only).
// ---------------------------------------------------------------------------------------------------
template <bool WITH_MAX, bool VEC>
__global__ __launch_bounds__(256) void pool_rows_kernel(const float* __restrict__ x, float* __restrict__ avg,
                                                       float* __restrict__ mx, int C, int HW, int groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * RPB;
    for (int r = wave; r < RPB && c0 + r < C; r += 4) {
        const long row = (long)b * C + c0 + r;
        const float* p = x + row * HW;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        float m = -INFINITY;
        if constexpr (VEC) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            const int n4 = HW >> 2;
            int i = lane;
            for (; i + 192 < n4; i += 256) {
                float4 a = p4[i], b = p4[i + 64], c = p4[i + 128], d = p4[i + 192];
                s0 += (a.x + b.x) + (c.x + d.x);
                s1 += (a.y + b.y) + (c.y + d.y);
                s2 += (a.z + b.z) + (c.z + d.z);
                s3 += (a.w + b.w) + (c.w + d.w);
                if constexpr (WITH_MAX) {
                    m = fmaxf(m, fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
                    m = fmaxf(m, fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
                }
            }
            for (; i < n4; i += 64) {
                float4 a = p4[i];
                s0 += a.x; s1 += a.y; s2 += a.z; s3 += a.w;
                if constexpr (WITH_MAX) m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
            }
        } else {
            for (int i = lane; i < HW; i += 64) {
                float a = p[i];
                s0 += a;
                if constexpr (WITH_MAX) m = fmaxf(m, a);
            }
        }
        float s = wave_sum((s0 + s1) + (s2 + s3));
        if constexpr (WITH_MAX) m = wave_max(m);
        if (lane == 0) {
            avg[row] = s / (float)HW;
            if constexpr (WITH_MAX) mx[row] = m;
        }
    }
}

// Second-pass block order.  With `reverse` the pass walks the (image, group) list backwards in units of 8 so
// that (a) the rows the previous pass touched LAST (still in L2 / Infinity Cache) are re-read FIRST and (b) a
// group stays on the XCD (block id mod 8) that pooled it.
__device__ __forceinline__ int second_pass_block(int bid, int nblocks, int reverse) {
    if (!reverse) return bid;
    const int full = nblocks & ~7;                       // reversible part; the (<8) tail keeps its place
    if (bid >= full) return bid;
    return (full - 8 - (bid & ~7)) + (bid & 7);
}

// ---------------------------------------------------------------------------------------------------
// Excite helpers (run by one whole 256-thread workgroup; smem: p[C] | h[Cr]).
// hidden_j = relu(sum_c w1[j,c] * p[c]) computed by 16-lane groups (16 j's per pass).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot16(const float* __restrict__ wrow, const float* s_p, int C, int part) {
    float acc = 0.f;
    for (int c = part; c < C; c += 16) acc += wrow[c] * s_p[c];
    acc += __shfl_xor(acc, 8, WAVE);
    acc += __shfl_xor(acc, 4, WAVE);
    acc += __shfl_xor(acc, 2, WAVE);
    acc += __shfl_xor(acc, 1, WAVE);
    return acc;
}

// ---------------------------------------------------------------------------------------------------
// K2 (SE / ECA): recompute the gates of this workgroup's RPB channels from the pooled vector, then
// stream y = x * g.  grid = B * ceil(C / RPB).
//   MODE 0: SE   gate_c = sigmoid(sum_j w2[c,j] relu(sum_c' w1[j,c'] p[c']))
//   MODE 1: ECA  gate_c = sigmoid(sum_j wk[j] p[c + j - pad])
// ---------------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ v4f ldx(const v4f* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void stx(v4f v, v4f* p) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int MODE, bool VEC, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void gate_scale_kernel(const float* __restrict__ x, const float* __restrict__ pooled,
                                                        const float* __restrict__ wa, const float* __restrict__ wb,
                                                        float* __restrict__ y, int C, int Cr, int HW, int groups, int reverse,
                                                        const mi355::SeExtra ex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_p = smem;            // C
    float* s_h = smem + C;        // Cr (SE)
    float* s_g = s_h + Cr;        // RPB
    const int t = threadIdx.x;
    const int blk = second_pass_block(blockIdx.x, gridDim.x, reverse);
    const int b = blk / groups;
    const int c0 = (blk % groups) * RPB;
    const float* pb = pooled + (long)b * C;

    if constexpr (MODE == 0) {
        for (int c = t; c < C; c += 256) s_p[c] = pb[c];
        __syncthreads();
        const int part = t & 15, jl = t >> 4;
        for (int j0 = 0; j0 < Cr; j0 += 16) {
            const int j = j0 + jl;
            float acc = (j < Cr) ? dot16(wa + (long)j * C, s_p, C, part) : 0.f;
            if (part == 0 && j < Cr) s_h[j] = relu_nan(acc + (ex.b1 ? ex.b1[j] : 0.f));
        }
        __syncthreads();
        if (t < RPB && c0 + t < C) {
            const float* w2r = wb + (long)(c0 + t) * Cr;
            float z = 0.f;
            for (int j = 0; j < Cr; ++j) z += w2r[j] * s_h[j];
            s_g[t] = se_gate(z + (ex.b2 ? ex.b2[c0 + t] : 0.f), ex.gate);
        }
    } else {
        if (t < RPB && c0 + t < C) {
            const int k = Cr, pad = (k - 1) / 2, c = c0 + t;
            float z = 0.f;
            for (int j = 0; j < k; ++j) {
                const int cc = c + j - pad;
                if (cc >= 0 && cc < C) z += wa[j] * pb[cc];
            }
            s_g[t] = sigmoidf_(z);
        }
    }
    __syncthreads();

    const int lane = t & 63, wave = t >> 6;
    for (int r = wave; r < RPB && c0 + r < C; r += 4) {
        const float g = s_g[r];
        const long off = ((long)b * C + c0 + r) * HW;
        if constexpr (VEC) {
            const v4f* xr = reinterpret_cast<const v4f*>(x + off);
            v4f* yr = reinterpret_cast<v4f*>(y + off);
            const int n4 = HW >> 2;
            int i = lane;
            for (; i + 192 < n4; i += 256) {
                v4f a = ldx<NTL>(&xr[i]), bq = ldx<NTL>(&xr[i + 64]);
                v4f c = ldx<NTL>(&xr[i + 128]), d = ldx<NTL>(&xr[i + 192]);
                stx<NTS>(a * g, &yr[i]);
                stx<NTS>(bq * g, &yr[i + 64]);
                stx<NTS>(c * g, &yr[i + 128]);
                stx<NTS>(d * g, &yr[i + 192]);
            }
            for (; i < n4; i += 64) {
                v4f a = ldx<NTL>(&xr[i]);
                stx<NTS>(a * g, &yr[i]);
            }
        } else {
            for (int i = lane; i < HW; i += 64) y[off + i] = x[off + i] * g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// CBAM channel gates of one image, computed cooperatively by a 256-thread workgroup into LDS:
//   gc[c] = sigmoid(W2 (relu(W1 avg) + relu(W1 max)))  (cbam.py:31-35; fc has no bias, so fc(avg)+fc(max)
//   collapses to one W2 product).  12K MACs -- cheaper to recompute in every consumer workgroup than to
//   launch a kernel for it.  scratch: a[C] | m[C] | h[Cr];  result: s_gc[C].  Ends with a barrier.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cbam_channel_gates(const float* __restrict__ avg_b, const float* __restrict__ mx_b,
                                                   const float* __restrict__ w1, const float* __restrict__ w2, int C, int Cr,
                                                   float* scratch, float* s_gc) {
    float* s_a = scratch;
    float* s_m = scratch + C;
    float* s_h = s_m + C;
    const int t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        s_a[c] = avg_b[c];
        s_m[c] = mx_b[c];
    }
    __syncthreads();
    const int part = t & 15, jl = t >> 4;
    for (int j0 = 0; j0 < Cr; j0 += 16) {
        const int j = j0 + jl;
        float ha = 0.f, hm = 0.f;
        if (j < Cr) {
            ha = dot16(w1 + (long)j * C, s_a, C, part);
            hm = dot16(w1 + (long)j * C, s_m, C, part);
        }
        if (part == 0 && j < Cr) s_h[j] = relu_nan(ha) + relu_nan(hm);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        const float* w2r = w2 + (long)c * Cr;
        float z = 0.f;
        for (int j = 0; j < Cr; ++j) z += w2r[j] * s_h[j];
        s_gc[c] = sigmoidf_(z);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// CBAM K3: per-pixel statistics over channels of x' = x * gc:  smap[b,0,p] = mean_c x', smap[b,1,p] = max_c x'
// (order [avg,max], cbam.py:46).  Workgroup = (image, tile of 64 lanes x VW pixels); the 4 waves split the
// channel range and combine through LDS.  gc may be NULL (stage 2: SpatialAttention alone).
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void cbam_spatial_stats_kernel(const float* __restrict__ x, const float* __restrict__ avg,
                                                                const float* __restrict__ mx, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ smap, int C,
                                                                int Cr, int HW, int tiles) {
    constexpr int VW = VEC ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // gc[C] | scratch a[C] m[C] h[Cr]   (only when avg != null)
    __shared__ __attribute__((aligned(16))) float s_sum[4][64 * VW];
    __shared__ __attribute__((aligned(16))) float s_max[4][64 * VW];
    const bool has_c = avg != nullptr;
    if (has_c) {
        const int bb = blockIdx.x / tiles;
        cbam_channel_gates(avg + (long)bb * C, mx + (long)bb * C, w1, w2, C, Cr, smem + C, smem);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int nv = HW / VW;                       // vector elements per row
    const int iv = tile * 64 + lane;
    const bool ok = iv < nv;
    float s[VW], m[VW];
#pragma unroll
    for (int q = 0; q < VW; ++q) { s[q] = 0.f; m[q] = -INFINITY; }
    const float* xb = x + (long)b * C * HW;
    if (ok) {
#pragma unroll 8
        for (int c = wave; c < C; c += 4) {
            const float g = has_c ? smem[c] : 1.0f;
            if constexpr (VEC) {
                const float4 v = reinterpret_cast<const float4*>(xb + (long)c * HW)[iv];
                const float a0 = v.x * g, a1 = v.y * g, a2 = v.z * g, a3 = v.w * g;
                s[0] += a0; s[1] += a1; s[2] += a2; s[3] += a3;
                m[0] = fmaxf(m[0], a0); m[1] = fmaxf(m[1], a1); m[2] = fmaxf(m[2], a2); m[3] = fmaxf(m[3], a3);
            } else {
                const float a0 = xb[(long)c * HW + iv] * g;
                s[0] += a0; m[0] = fmaxf(m[0], a0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < VW; ++q) { s_sum[wave][lane * VW + q] = s[q]; s_max[wave][lane * VW + q] = m[q]; }
    __syncthreads();
    // 64*VW outputs per tile, combined by the first 64*VW threads
    const int o = threadIdx.x;
    if (o < 64 * VW) {
        const int pix = tile * 64 * VW + o;
        if (pix < HW) {
            const float ss = (s_sum[0][o] + s_sum[1][o]) + (s_sum[2][o] + s_sum[3][o]);
            const float mm = fmaxf(fmaxf(s_max[0][o], s_max[1][o]), fmaxf(s_max[2][o], s_max[3][o]));
            smap[((long)b * 2 + 0) * HW + pix] = ss / (float)C;
            smap[((long)b * 2 + 1) * HW + pix] = mm;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// CBAM K4: y = (x * gc[b,c]) * gs[b,p] for one (image, band of TR rows) -- all channels.  The workgroup first rebuilds
// what it needs on chip: the channel gates (see above), then the spatial gate of ITS band,
//   gs = sigmoid(conv_ks x ks(smap[b]))  (2->1 channels, zero pad ks/2, no bias, cross-correlation; cbam.py:43-48),
// from an LDS tile of the 2-channel map (band + halo, 4 outputs per thread), so neither gate ever goes through HBM
// and no gate kernel sits between the streaming passes.  Then it streams the band of every channel (rounding order
// of the reference: the channel-stage product is rounded to fp32 before the spatial multiply).
// smem: gc[C] | scratch[2C+Cr] | w[2*ks*ks (pad 4)] | tile[2][TH][TW] | gs[TR*W (pad 4)]
// ---------------------------------------------------------------------------------------------------
template <int KS, bool VEC, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void cbam_apply_band_kernel(const float* __restrict__ x, const float* __restrict__ avg,
                                                             const float* __restrict__ mx, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, const float* __restrict__ smap,
                                                             const float* __restrict__ wconv, float* __restrict__ y, int C, int Cr,
                                                             int H, int W, int ks_rt, int TR, int bands) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool has_c = avg != nullptr, has_s = smap != nullptr;
    const int ks = KS > 0 ? KS : ks_rt;
    const int kk = ks * ks, pad = ks / 2;
    const int quads = (W + 3) >> 2;
    const int TW = quads * 4 + ks - 1, TH = TR + ks - 1;
    float* s_gc = smem;
    float* s_scr = smem + C;
    float* s_w = smem + ((3 * C + Cr + 3) & ~3);             // keep the float4-read regions 16-byte aligned
    float* s_t = s_w + ((2 * kk + 3) & ~3);
    float* s_gs = s_t + ((2 * TH * TW + 3) & ~3);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x / bands, r0 = (blockIdx.x % bands) * TR;
    const int rows_here = min(TR, H - r0);
    const int HW = H * W;

    if (has_c) cbam_channel_gates(avg + (long)b * C, mx + (long)b * C, w1, w2, C, Cr, s_scr, s_gc);
    if (has_s) {
        for (int i = t; i < 2 * kk; i += 256) s_w[i] = wconv[i];
        const float* sb = smap + (long)b * 2 * HW;
        for (int i = t; i < 2 * TH * TW; i += 256) {
            const int ch = i / (TH * TW), rem = i % (TH * TW);
            const int ty = rem / TW, tx = rem % TW;
            const int gy = r0 + ty - pad, gx = tx - pad;
            s_t[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? sb[((long)ch * H + gy) * W + gx] : 0.f;
        }
        __syncthreads();
        for (int o = t; o < rows_here * quads; o += 256) {
            const int r = o / quads, col = (o % quads) * 4;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int ch = 0; ch < 2; ++ch) {
                if constexpr (KS > 0) {
#pragma unroll
                    for (int dy = 0; dy < KS; ++dy) {
                        const float* trow = s_t + (ch * TH + r + dy) * TW + col;
                        const float* wrow = s_w + ch * kk + dy * KS;
                        float v[KS + 3];
#pragma unroll
                        for (int i = 0; i < KS + 3; ++i) v[i] = trow[i];
#pragma unroll
                        for (int dx = 0; dx < KS; ++dx) {
                            const float w = wrow[dx];
                            a0 += w * v[dx]; a1 += w * v[dx + 1]; a2 += w * v[dx + 2]; a3 += w * v[dx + 3];
                        }
                    }
                } else {
                    for (int dy = 0; dy < ks; ++dy) {
                        const float* trow = s_t + (ch * TH + r + dy) * TW + col;
                        const float* wrow = s_w + ch * kk + dy * ks;
                        for (int dx = 0; dx < ks; ++dx) {
                            const float w = wrow[dx];
                            a0 += w * trow[dx]; a1 += w * trow[dx + 1]; a2 += w * trow[dx + 2]; a3 += w * trow[dx + 3];
                        }
                    }
                }
            }
            const float res[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col + q < W) s_gs[r * W + col + q] = sigmoidf_(res[q]);
        }
    }
    __syncthreads();

    // ---- stream the band of every channel: waves take channels round-robin, (channel, vector) pairs are walked flat so
    //      that all 64 lanes stay busy whatever the band length --------------------------------------------------------------
    const long band0 = (long)r0 * W;
    const int npx = rows_here * W;
    const float* xb = x + (long)b * C * HW + band0;
    float* yb = y + (long)b * C * HW + band0;
    if constexpr (VEC) {
        const int n4 = npx >> 2;                                    // W % 4 == 0 on this path
        int c = wave, i = lane;
        while (i >= n4) { i -= n4; c += 4; }
        while (c < C) {
            const float g = has_c ? s_gc[c] : 1.0f;
            v4f s4 = {1.f, 1.f, 1.f, 1.f};
            if (has_s) s4 = reinterpret_cast<const v4f*>(s_gs)[i];
            v4f v = ldx<NTL>(&reinterpret_cast<const v4f*>(xb + (long)c * HW)[i]);
            v = (v * g) * s4;
            stx<NTS>(v, &reinterpret_cast<v4f*>(yb + (long)c * HW)[i]);
            i += 64;
            while (i >= n4) { i -= n4; c += 4; }
        }
    } else {
        for (int c = wave; c < C; c += 4) {
            const float g = has_c ? s_gc[c] : 1.0f;
            for (int i = lane; i < npx; i += 64) {
                const float sp = has_s ? s_gs[i] : 1.0f;
                yb[(long)c * HW + i] = (xb[(long)c * HW + i] * g) * sp;
            }
        }
    }
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// read-only sweep (sum reduction, one atomic-free store per block only if the sum is NaN-free-impossible):
// the bandwidth yardstick for the pooling passes and for Infinity-Cache residency experiments.
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ src, long n4, float* __restrict__ sink) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s0 += a.x + b.x + c.x + d.x; s1 += a.y + b.y + c.y + d.y;
        s2 += a.z + b.z + c.z + d.z; s3 += a.w + b.w + c.w + d.w;
    }
    for (; i < n4; i += stride) { float4 a = src[i]; s0 += a.x; s1 += a.y; s2 += a.z; s3 += a.w; }
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0 && s == 123456.789f) sink[0] = s;   // keeps the loads live; practically never stores
}

struct Tune { int chunk; int ntl; int nts; int reverse; };

inline Tune resolve_tune(int B, long bytes_per_image) {
    Tune t;
    long ci = mi355::opt_chunk_images();
    if (ci <= 0) {   // auto: ~200 MB of x per chunk, so the chunk's re-read(s) are served by the 256 MiB Infinity Cache
        ci = (200L << 20) / (bytes_per_image > 0 ? bytes_per_image : 1);
        if (ci < 1) ci = 1;
    }
    t.chunk = ci > B ? B : (int)ci;
    const long nt = mi355::opt_nt();
    t.ntl = (nt & 1) != 0;
    t.nts = (nt & 2) != 0;
    t.reverse = mi355::opt_reverse() != 0;
    return t;
}

#define NT_DISPATCH(ntl, nts, CALL)                                        \
    do {                                                                   \
        if (ntl) { if (nts) { CALL(true, true); } else { CALL(true, false); } } \
        else     { if (nts) { CALL(false, true); } else { CALL(false, false); } } \
    } while (0)

}  // namespace

// ===================================================================================================
// C ABI
// ===================================================================================================
extern "C" {

// workspace: pooled means [B*C] (rounded to 16 B) | single-pass sync state (arrive[B], ticket, err)
static inline size_t pooled_bytes(int B, int C) { return (((size_t)B * C * sizeof(float)) + 15) & ~(size_t)15; }
size_t mi355_se_workspace_bytes(int B, int C, int, int) {
    return pooled_bytes(B, C) + mi355::fused_state_bytes(B) + mi355::se_single_extra_bytes(B, C);
}
size_t mi355_eca_workspace_bytes(int B, int C, int, int) { return pooled_bytes(B, C) + mi355::fused_state_bytes(B); }

static int se_eca_common(int mode, const float* x, const float* wa, const float* wb, float* y, int B, int C, int Cr,
                         int H, int W, void* ws, size_t ws_bytes, hipStream_t st, mi355::SeExtra ex = mi355::SeExtra{nullptr, nullptr, 0}) {
    const int HW = H * W;
    const bool vec = (HW % 4 == 0) && aligned16(x) && aligned16(y);
    float* pooled = static_cast<float*>(ws);
    const int groups = cdiv(C, RPB);
    const size_t smem = (size_t)(C + (mode == 0 ? Cr : 0) + RPB) * sizeof(float);
    if (smem > 64 * 1024) return mi355::fail(MI355_EUNSUPPORTED, "channel count %d too large for the gate stage", C);
    if (mode == 0 && vec && mi355::se_single_applicable(C, Cr, H, W)) {       // SE: x read once, means exchanged as tagged granules
        char* state = static_cast<char*>(ws) + pooled_bytes(B, C);
        return mi355::se_single(x, wa, wb, y, B, C, Cr, H, W, state, state + mi355::fused_state_bytes(B), ex, st);
    }
    if (mode == 1 && vec && mi355::eca_single_applicable(C, Cr, H, W))         // ECA: x read once, no exchange between workgroups
        return mi355::eca_single(x, wa, y, B, C, Cr, H, W, st);
    const Tune tu = resolve_tune(B, (long)C * HW * 4);
    for (int b0 = 0; b0 < B; b0 += tu.chunk) {
        const int nb = (B - b0 < tu.chunk) ? B - b0 : tu.chunk;
        const float* xc = x + (long)b0 * C * HW;
        float* yc = y + (long)b0 * C * HW;
        float* pc = pooled + (long)b0 * C;
        const int grid = nb * groups;
        if (vec) pool_rows_kernel<false, true><<<grid, 256, 0, st>>>(xc, pc, nullptr, C, HW, groups);
        else     pool_rows_kernel<false, false><<<grid, 256, 0, st>>>(xc, pc, nullptr, C, HW, groups);
#define SCALE_CALL(NTL, NTS)                                                                                          \
        do {                                                                                                           \
            if (mode == 0) {                                                                                           \
                if (vec) gate_scale_kernel<0, true, NTL, NTS><<<grid, 256, smem, st>>>(xc, pc, wa, wb, yc, C, Cr, HW, groups, tu.reverse, ex);   \
                else     gate_scale_kernel<0, false, NTL, NTS><<<grid, 256, smem, st>>>(xc, pc, wa, wb, yc, C, Cr, HW, groups, tu.reverse, ex);  \
            } else {                                                                                                   \
                if (vec) gate_scale_kernel<1, true, NTL, NTS><<<grid, 256, smem, st>>>(xc, pc, wa, nullptr, yc, C, Cr, HW, groups, tu.reverse, ex);  \
                else     gate_scale_kernel<1, false, NTL, NTS><<<grid, 256, smem, st>>>(xc, pc, wa, nullptr, yc, C, Cr, HW, groups, tu.reverse, ex); \
            }                                                                                                          \
        } while (0)
        NT_DISPATCH(tu.ntl, tu.nts, SCALE_CALL);
#undef SCALE_CALL
    }
    MI355_LAUNCH_CHECK();
    (void)ws_bytes;
    return MI355_OK;
}

int mi355_se_fwd(const float* x, const float* w1, const float* w2, float* y, int B, int C, int Cr, int H, int W,
                 void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w1 && w2 && y && ws);
    MI355_CHECK_ARG(B > 0 && C > 0 && Cr > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(ws_bytes >= mi355_se_workspace_bytes(B, C, H, W));
    return se_eca_common(0, x, w1, w2, y, B, C, Cr, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mi355_se_ex_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int B, int C, int Cr,
                    int H, int W, int gate, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w1 && w2 && y && ws);
    MI355_CHECK_ARG(B > 0 && C > 0 && Cr > 0 && H > 0 && W > 0 && (gate == 0 || gate == 1));
    MI355_CHECK_ARG(ws_bytes >= mi355_se_workspace_bytes(B, C, H, W));
    return se_eca_common(0, x, w1, w2, y, B, C, Cr, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream), mi355::SeExtra{b1, b2, gate});
}

int mi355_eca_fwd(const float* x, const float* wconv, float* y, int B, int C, int k, int H, int W, void* ws,
                  size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && wconv && y && ws);
    MI355_CHECK_ARG(B > 0 && C > 0 && k > 0 && (k & 1) && H > 0 && W > 0);
    MI355_CHECK_ARG(ws_bytes >= mi355_eca_workspace_bytes(B, C, H, W));
    return se_eca_common(1, x, wconv, nullptr, y, B, C, k, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

// workspace: avg[B*C] | max[B*C] | smap[B*2*HW]   (each region rounded to 16 B)
static inline size_t r16(size_t n) { return (n + 15) & ~(size_t)15; }
static inline size_t cbam_multipass_bytes(int B, int C, int H, int W) {
    return 2 * r16((size_t)B * C * 4) + r16((size_t)B * 2 * H * W * 4);
}
size_t mi355_cbam_workspace_bytes(int B, int C, int H, int W) {
    return cbam_multipass_bytes(B, C, H, W) + mi355::cbam_single_extra_bytes(B, C, H, W);
}

int mi355_cbam_fwd(const float* x, const float* w1, const float* w2, const float* wconv, float* y, int B, int C, int Cr,
                   int ks, int H, int W, int stage, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && ws);
    MI355_CHECK_ARG(stage >= 0 && stage <= 2);
    MI355_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0);
    const bool do_c = stage != 2, do_s = stage != 1;
    if (do_c) MI355_CHECK_ARG(w1 && w2 && Cr > 0);
    if (do_s) MI355_CHECK_ARG(wconv && ks > 0 && (ks & 1));
    MI355_CHECK_ARG(ws_bytes >= mi355_cbam_workspace_bytes(B, C, H, W));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stage == 0 && aligned16(x) && aligned16(y) && mi355::cbam_single_applicable(C, Cr, H, W, ks))                                                   // x read once
        return mi355::cbam_single(x, w1, w2, wconv, y, B, C, Cr, H, W, ks, static_cast<char*>(ws) + cbam_multipass_bytes(B, C, H, W), st);
    if (!do_c) Cr = 0;
    if (!do_s) ks = 1;
    const int HW = H * W;
    const bool vec_pix = (HW % 4 == 0) && aligned16(x) && aligned16(y);          // whole-image rows
    const bool vec_band = vec_pix && (W % 4 == 0);                               // row bands start 16-byte aligned
    char* wsp = static_cast<char*>(ws);
    const size_t bc = r16((size_t)B * C * 4);
    float* avg = reinterpret_cast<float*>(wsp);
    float* mx = reinterpret_cast<float*>(wsp + bc);
    float* smap = reinterpret_cast<float*>(wsp + 2 * bc);

    // band geometry of the final pass: ~8 rows per workgroup (>= 1 output quad per thread for the gate conv), LDS <= 60 KiB
    const int quads = (W + 3) / 4, TW = quads * 4 + ks - 1, wpad = (2 * ks * ks + 3) & ~3;
    const size_t gate_floats = (size_t)((3 * C + Cr + 3) & ~3);
    int TR = 256 / quads;
    if (TR > 8) TR = 8;
    if (TR < 1) TR = 1;
    if (TR > H) TR = H;
    auto apply_smem = [&](int tr) {
        return (gate_floats + wpad + (size_t)((2 * (tr + ks - 1) * TW + 3) & ~3) + (size_t)((tr * W + 3) & ~3)) * 4;
    };
    while (TR > 1 && apply_smem(TR) > 60 * 1024) --TR;
    if (apply_smem(TR) > 60 * 1024)
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_cbam_fwd: C=%d / W=%d exceed the LDS budget of the fused final pass", C, W);
    const int bands = cdiv(H, TR);
    const size_t smem_apply = apply_smem(TR);
    const size_t smem_stats = do_c ? gate_floats * 4 : 0;

    const int groups = cdiv(C, RPB);
    const int VW = vec_pix ? 4 : 1;
    const int tiles = cdiv(HW / VW, 64);
    const Tune tu = resolve_tune(B, (long)C * HW * 4);
    for (int b0 = 0; b0 < B; b0 += tu.chunk) {
        const int nb = (B - b0 < tu.chunk) ? B - b0 : tu.chunk;
        const float* xc = x + (long)b0 * C * HW;
        float* yc = y + (long)b0 * C * HW;
        float* avgc = do_c ? avg + (long)b0 * C : nullptr;
        float* mxc = do_c ? mx + (long)b0 * C : nullptr;
        float* smc = do_s ? smap + (long)b0 * 2 * HW : nullptr;
        if (do_c) {
            if (vec_pix) pool_rows_kernel<true, true><<<nb * groups, 256, 0, st>>>(xc, avgc, mxc, C, HW, groups);
            else         pool_rows_kernel<true, false><<<nb * groups, 256, 0, st>>>(xc, avgc, mxc, C, HW, groups);
        }
        if (do_s) {
            if (vec_pix) cbam_spatial_stats_kernel<true><<<nb * tiles, 256, smem_stats, st>>>(xc, avgc, mxc, w1, w2, smc, C, Cr, HW, tiles);
            else         cbam_spatial_stats_kernel<false><<<nb * tiles, 256, smem_stats, st>>>(xc, avgc, mxc, w1, w2, smc, C, Cr, HW, tiles);
        }
#define APPLY_KS(KS_, NTL, NTS)                                                                                                   \
        do {                                                                                                                       \
            if (vec_band) cbam_apply_band_kernel<KS_, true, NTL, NTS><<<nb * bands, 256, smem_apply, st>>>(xc, avgc, mxc, w1, w2, smc, wconv, yc, C, Cr, H, W, ks, TR, bands);  \
            else          cbam_apply_band_kernel<KS_, false, NTL, NTS><<<nb * bands, 256, smem_apply, st>>>(xc, avgc, mxc, w1, w2, smc, wconv, yc, C, Cr, H, W, ks, TR, bands); \
        } while (0)
#define APPLY_CALL(NTL, NTS)                                  \
        do {                                                   \
            if (ks == 7) APPLY_KS(7, NTL, NTS);                \
            else if (ks == 3) APPLY_KS(3, NTL, NTS);           \
            else APPLY_KS(0, NTL, NTS);                        \
        } while (0)
        NT_DISPATCH(tu.ntl, tu.nts, APPLY_CALL);
#undef APPLY_CALL
#undef APPLY_KS
    }
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_stream_copy(const void* src, void* dst, size_t bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(src && dst && bytes % 16 == 0 && aligned16(src) && aligned16(dst));
    const long n4 = (long)(bytes / 16);
    if (n4 == 0) return MI355_OK;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    MI355_TRACE(static_cast<hipStream_t>(stream), "stream_copy_kernel bytes=%zu", bytes);
    stream_copy_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const float4*>(src),
                                                                            static_cast<float4*>(dst), n4);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_stream_read(const void* src, size_t bytes, float* sink, mi355_stream_t stream) {
    MI355_CHECK_ARG(src && sink && bytes % 16 == 0 && aligned16(src));
    const long n4 = (long)(bytes / 16);
    if (n4 == 0) return MI355_OK;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    stream_read_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const float4*>(src), n4, sink);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
