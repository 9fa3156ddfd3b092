// axis_attn.hip -- four more members of the reference's attention zoo whose gate is built from AXIS reductions of the
// NCHW activation: GCModule (gc_module.py:17-43), CoordinateAttention (coordatten.py:18-44), TripletAttention
// (triplet_attention.py:19-63) and BAM (bam.py:16-71).  All are HBM-bound: x is read once per reduction family and once more by
// the apply pass, everything in between (pooled axes, tiny convolutions / MLPs, gate planes) is KB..MB sized and lives in the
// caller's workspace.  Shared building blocks:
//     chan_reduce_kernel   reductions over C for every pixel      (1x1 conv to K planes, or ZPool = mean & max over channels)
//     plane_pool_kernel    reductions over W and over H for every (image, channel) plane  (mean, optionally max)
//     plane_dot_kernel     sum_hw x[b,c,hw] * v[b,hw]             (GC context aggregation)
//     gate_conv_kernel     k x k convolution of a 2-plane map -> BatchNorm(eval, folded) -> ReLU -> sigmoid (Triplet's AttentionGate)
//     small_conv_kernel    dilated 3x3 convolution on a handful of channels + folded BatchNorm + ReLU (BAM's spatial gate)
//     apply_kernel<MODE>   the broadcast pass
#include "common.h"
#include "mma.h"

namespace {

enum { AP_GC = 0, AP_COORD = 1, AP_TRIPLET = 2, AP_BAM = 3 };

// ---- reductions over the channel axis ---------------------------------------------------------------------------------------
// MODE 0: out[b, k, p] = bias[k] + sum_c w[k*C + c] * x[b, c, p]  for k < K (K <= KMAX)
// MODE 1: out[b, 0, p] = mean_c x[b, c, p],  out[b, 1, p] = max_c x[b, c, p]
// A thread owns VEC consecutive pixels; accumulators are kept as vector values (a 2-D scalar array of this size is left in scratch
// by the compiler).  Ragged C: channels past the end load the last channel again with weight 0 (-inf for the max).
template <int VEC> struct PixVec;
template <> struct PixVec<4> { using t = f4; };
template <> struct PixVec<1> { using t = float; };
__device__ __forceinline__ f4 vmaxf(f4 a, f4 b) { return f4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
__device__ __forceinline__ float vmaxf(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f4 splat(f4, float s) { return f4{s, s, s, s}; }
__device__ __forceinline__ float splat(float, float s) { return s; }

template <int MODE, int KMAX, int VEC>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int C, long HW, int K) {
    using V = typename PixVec<VEC>::t;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= HW) return;
    const int b = blockIdx.y;
    const float* xp = x + (long)b * C * HW + p;
    constexpr int NA = MODE == 0 ? KMAX : 2;
    V acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = splat(V{}, (MODE == 1 && k == 1) ? -INFINITY : 0.f);
    constexpr int U = (MODE == 0 && KMAX >= 16) ? 4 : 8;
    for (int c = 0; c < C; c += U) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const V*>(xp + (long)(c + u < C ? c + u : C - 1) * HW);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = c + u < C;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const float wk = (k < K && in) ? w[(long)k * C + c + u] : 0.f;
                    acc[k] += wk * v[u];
                }
            } else {
                acc[0] += in ? v[u] : splat(V{}, 0.f);
                acc[1] = vmaxf(acc[1], in ? v[u] : splat(V{}, -INFINITY));
            }
        }
    }
    const int nout = MODE == 0 ? K : 2;
    float* op = out + (long)b * nout * HW + p;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        if (k < nout) {
            V r;
            if constexpr (MODE == 0) r = acc[k] + splat(V{}, bias ? bias[k] : 0.f);
            else                     r = k == 0 ? acc[0] / (float)C : acc[1];
            *reinterpret_cast<V*>(op + (long)k * HW) = r;
        }
    }
}

template <int MODE, int KMAX>
void launch_chan_reduce(const float* x, const float* w, const float* bias, float* out, int B, int C, long HW, int K, hipStream_t st) {
    const bool vec = (HW & 3) == 0 && aligned16(x) && aligned16(out);
    if (vec) chan_reduce_kernel<MODE, KMAX, 4><<<dim3(cdiv(HW / 4, 256), B), 256, 0, st>>>(x, w, bias, out, C, HW, K);
    else     chan_reduce_kernel<MODE, KMAX, 1><<<dim3(cdiv(HW, 256), B), 256, 0, st>>>(x, w, bias, out, C, HW, K);
}

// ---- reductions over W (one value per row) and over H (one value per column) of every (image, channel) plane -----------------
// One wave per plane, lanes along a row (NCH chunks of 64 columns), eight rows of loads in flight.  Column results live in the
// lanes; a row's wave-reduced result is parked in lane (row & 63) and flushed as one coalesced store per 64 rows.
template <bool WITH_MAX, int NCH>
__global__ __launch_bounds__(256) void plane_pool_kernel(const float* __restrict__ x, float* __restrict__ h_mean, float* __restrict__ h_max,
                                                        float* __restrict__ w_mean, float* __restrict__ w_max, long planes, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* xp = x + plane * (long)H * W;
    float csum[NCH], cmax[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { csum[ch] = 0.f; cmax[ch] = -INFINITY; }
    float keep_s = 0.f, keep_m = 0.f;
    const float inv_w = 1.f / (float)W, inv_h = 1.f / (float)H;
    constexpr int RU = NCH == 1 ? 8 : 4;
    for (int i0 = 0; i0 < H; i0 += RU) {
        float v[RU][NCH];
#pragma unroll
        for (int r = 0; r < RU; ++r)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int j = ch * 64 + lane;
                v[r][ch] = (i0 + r < H && j < W) ? xp[(long)(i0 + r) * W + j] : 0.f;
            }
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            const int i = i0 + r;
            if (i >= H) break;
            float rs = 0.f, rm = -INFINITY;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const bool in = ch * 64 + lane < W;
                csum[ch] += v[r][ch];
                rs += v[r][ch];
                if constexpr (WITH_MAX) {
                    const float vm = in ? v[r][ch] : -INFINITY;
                    cmax[ch] = fmaxf(cmax[ch], vm);
                    rm = fmaxf(rm, vm);
                }
            }
            rs = wave_sum(rs);
            if constexpr (WITH_MAX) rm = wave_max(rm);
            if (lane == (i & 63)) { keep_s = rs * inv_w; keep_m = rm; }
            if ((i & 63) == 63 || i == H - 1) {
                const int base = i & ~63;
                if (base + lane <= i) {
                    h_mean[plane * H + base + lane] = keep_s;
                    if constexpr (WITH_MAX) h_max[plane * H + base + lane] = keep_m;
                }
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch * 64 + lane;
        if (j < W) {
            w_mean[plane * W + j] = csum[ch] * inv_h;
            if constexpr (WITH_MAX) w_max[plane * W + j] = cmax[ch];
        }
    }
}

template <bool WITH_MAX>
int launch_plane_pool(const float* x, float* h_mean, float* h_max, float* w_mean, float* w_max, long planes, int H, int W, hipStream_t st) {
    const int grid = cdiv(planes, 4);
    if (W <= 64)       plane_pool_kernel<WITH_MAX, 1><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else if (W <= 128) plane_pool_kernel<WITH_MAX, 2><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else if (W <= 256) plane_pool_kernel<WITH_MAX, 4><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else return mi355::fail(MI355_EUNSUPPORTED, "axis pooling: W = %d > 256", W);
    return MI355_OK;
}

// ---- out[plane] = scale * sum_p x[plane, p] * v[b, p]  (v == nullptr: plain sum) -- one wave per plane -------------------------
template <int VEC>
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ x, const float* __restrict__ v, float* __restrict__ out, long planes,
                                                       int C, long HW, float scale) {
    const int lane = threadIdx.x & 63;
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* xp = x + plane * HW;
    const float* vp = v ? v + (plane / C) * HW : nullptr;
    float s0 = 0.f, s1 = 0.f;
    if constexpr (VEC == 4) {
        const long n4 = HW >> 2;
        long q = lane;
        for (; q + 64 < n4; q += 128) {
            const f4 a = *reinterpret_cast<const f4*>(xp + q * 4), b = *reinterpret_cast<const f4*>(xp + (q + 64) * 4);
            if (vp) {
                const f4 u = *reinterpret_cast<const f4*>(vp + q * 4), w = *reinterpret_cast<const f4*>(vp + (q + 64) * 4);
                s0 += a.x * u.x + a.y * u.y + a.z * u.z + a.w * u.w;
                s1 += b.x * w.x + b.y * w.y + b.z * w.z + b.w * w.w;
            } else {
                s0 += (a.x + a.y) + (a.z + a.w);
                s1 += (b.x + b.y) + (b.z + b.w);
            }
        }
        for (; q < n4; q += 64) {
            const f4 a = *reinterpret_cast<const f4*>(xp + q * 4);
            if (vp) { const f4 u = *reinterpret_cast<const f4*>(vp + q * 4); s0 += a.x * u.x + a.y * u.y + a.z * u.z + a.w * u.w; }
            else s0 += (a.x + a.y) + (a.z + a.w);
        }
    } else {
        for (long q = lane; q < HW; q += 64) s0 += vp ? xp[q] * vp[q] : xp[q];
    }
    const float s = wave_sum(s0 + s1);
    if (lane == 0) out[plane] = s * scale;
}

void launch_plane_dot(const float* x, const float* v, float* out, long planes, int C, long HW, float scale, hipStream_t st) {
    const bool vec = (HW & 3) == 0 && aligned16(x) && (!v || aligned16(v));
    if (vec) plane_dot_kernel<4><<<cdiv(planes, 4), 256, 0, st>>>(x, v, out, planes, C, HW, scale);
    else     plane_dot_kernel<1><<<cdiv(planes, 4), 256, 0, st>>>(x, v, out, planes, C, HW, scale);
}

// ---- the broadcast pass ------------------------------------------------------------------------------------------------------
struct ApplyArgs {
    const float* x; float* y;
    const float* a;      // GC: t (B,C)            COORD: a_h (B,C,H)   TRIPLET: s_ch (B,C,H)   BAM: channel gate (B,C)
    const float* b;      //                        COORD: a_w (B,C,W)   TRIPLET: s_cw (B,C,W)   BAM: spatial gate (B,HW)
    const float* c;      //                                             TRIPLET: s_hw (B,HW)
    long total;          // B*C*HW / VEC
    int C, H, W;
};

template <int MODE, int VEC>
__global__ __launch_bounds__(256) void apply_kernel(const ApplyArgs g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.total) return;
    const long HW = (long)g.H * g.W;
    const long per = HW / VEC;
    const long plane = idx / per;
    const long p = (idx - plane * per) * VEC;
    const int i = (int)(p / g.W), j = (int)(p - (long)i * g.W);
    const long img = plane / g.C;
    float xv[VEC], yv[VEC];
    const float* xp = g.x + plane * HW + p;
    if constexpr (VEC == 4) { const f4 t = *reinterpret_cast<const f4*>(xp); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
    else xv[0] = xp[0];
    if constexpr (MODE == AP_GC) {
        const float t = g.a[plane];
#pragma unroll
        for (int e = 0; e < VEC; ++e) yv[e] = xv[e] + t;
    } else if constexpr (MODE == AP_COORD) {
        const float ah = g.a[plane * g.H + i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) yv[e] = xv[e] * ah * g.b[plane * g.W + j + e];
    } else if constexpr (MODE == AP_TRIPLET) {
        const float s1 = g.a[plane * g.H + i];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            yv[e] = (xv[e] * s1 + xv[e] * g.b[plane * g.W + j + e] + xv[e] * g.c[img * HW + p + e]) * (1.0f / 3.0f);
    } else {
        const float cg = g.a[plane];
#pragma unroll
        for (int e = 0; e < VEC; ++e) yv[e] = xv[e] + xv[e] * sigmoidf_(cg + g.b[img * HW + p + e]);
    }
    float* yp = g.y + plane * HW + p;
    if constexpr (VEC == 4) *reinterpret_cast<f4*>(yp) = f4{yv[0], yv[1], yv[2], yv[3]};
    else yp[0] = yv[0];
}

template <int MODE>
void launch_apply(ApplyArgs g, int B, hipStream_t st) {
    const long n = (long)B * g.C * g.H * g.W;
    const bool vec = (g.W & 3) == 0 && aligned16(g.x) && aligned16(g.y);
    if (vec) { g.total = n / 4; apply_kernel<MODE, 4><<<cdiv(g.total, 256), 256, 0, st>>>(g); }
    else     { g.total = n;     apply_kernel<MODE, 1><<<cdiv(g.total, 256), 256, 0, st>>>(g); }
}

// ---- GCModule: transform(context) = conv2( relu( LayerNorm_[Cr,1,1]( conv1(context) ) ) ), one workgroup per image ---------------
__global__ __launch_bounds__(256) void gc_transform_kernel(const float* __restrict__ ctx, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ tvec, int C, int Cr, float eps) {
    extern __shared__ float hbuf[];                              // Cr
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* cv = ctx + (long)b * C;
    for (int k = wave; k < Cr; k += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = __builtin_fmaf(w1[(long)k * C + c], cv[c], s);
        s = wave_sum(s);
        if (lane == 0) hbuf[k] = s + (b1 ? b1[k] : 0.f);
    }
    __syncthreads();
    float mean = 0.f;
    for (int k = 0; k < Cr; ++k) mean += hbuf[k];
    mean /= (float)Cr;
    float var = 0.f;
    for (int k = 0; k < Cr; ++k) { const float d = hbuf[k] - mean; var += d * d; }
    const float inv = 1.0f / sqrtf(var / (float)Cr + eps);
    for (int c = t; c < C; c += 256) {
        float s = b2 ? b2[c] : 0.f;
        for (int k = 0; k < Cr; ++k) {
            const float h = fmaxf((hbuf[k] - mean) * inv * ln_w[k] + ln_b[k], 0.f);
            s = __builtin_fmaf(w2[(long)c * Cr + k], h, s);
        }
        tvec[(long)b * C + c] = s;
    }
}

// ---- CoordinateAttention: y = relu(bn(conv1(cat[pool_h, pool_w]))) -> a_h = conv_h(y[:h]), a_w = conv_w(y[h:]) ------------------
// grid (ceil((H+W)/64), B); thread (k-slot = wave, position = lane).  bn folded: hid = relu(z * bn_s + bn_t).
__global__ __launch_bounds__(256) void coord_mlp_kernel(const float* __restrict__ ph, const float* __restrict__ pw, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                       const float* __restrict__ wh, const float* __restrict__ bh, const float* __restrict__ ww,
                                                       const float* __restrict__ bw, float* __restrict__ ah, float* __restrict__ aw, int C, int hid,
                                                       int H, int W) {
    extern __shared__ float hl[];                                // hid x 64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int pos = blockIdx.x * 64 + lane;
    const bool valid = pos < H + W, on_h = pos < H;
    const float* src = on_h ? ph + (long)b * C * H + pos : pw + (long)b * C * W + (pos - H);
    const int ld = on_h ? H : W;
    for (int k = wave; k < hid; k += 4) {
        float s = b1 ? b1[k] : 0.f;
        if (valid)
            for (int c = 0; c < C; ++c) s = __builtin_fmaf(w1[(long)k * C + c], src[(long)c * ld], s);
        hl[k * 64 + lane] = fmaxf(s * bn_s[k] + bn_t[k], 0.f);
    }
    __syncthreads();
    if (!valid) return;
    const float* wsel = on_h ? wh : ww;
    const float* bsel = on_h ? bh : bw;
    float* dst = on_h ? ah + (long)b * C * H + pos : aw + (long)b * C * W + (pos - H);
    for (int co = wave; co < C; co += 4) {
        float s = bsel ? bsel[co] : 0.f;
        for (int k = 0; k < hid; ++k) s = __builtin_fmaf(wsel[(long)co * hid + k], hl[k * 64 + lane], s);
        dst[(long)co * ld] = s;
    }
}

// ---- TripletAttention's AttentionGate on a pooled 2-plane map: sigmoid(relu(bn(conv_kxk([mean, max])))) ---------------------------
// in0 / in1: (B, R, S) planes with batch stride `bs`; w: (2, k, k); aff[0] = folded scale, aff[1] = folded shift (conv bias included).
__global__ __launch_bounds__(256) void gate_conv_kernel(const float* __restrict__ in0, const float* __restrict__ in1, long bs, const float* __restrict__ w,
                                                       const float* __restrict__ aff, float* __restrict__ out, int R, int S, int k) {
    __shared__ float wl[2 * 15 * 15];
    for (int q = threadIdx.x; q < 2 * k * k; q += 256) wl[q] = w[q];
    __syncthreads();
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long RS = (long)R * S;
    if (idx >= RS) return;
    const int b = blockIdx.y, r = (int)(idx / S), s = (int)(idx - (long)r * S), pad = (k - 1) / 2;
    const float* p0 = in0 + (long)b * bs;
    const float* p1 = in1 + (long)b * bs;
    float acc = 0.f;
    for (int u = 0; u < k; ++u) {
        const int rr = r + u - pad;
        if (rr < 0 || rr >= R) continue;
        for (int v = 0; v < k; ++v) {
            const int ss = s + v - pad;
            if (ss < 0 || ss >= S) continue;
            acc = __builtin_fmaf(wl[u * k + v], p0[(long)rr * S + ss], acc);
            acc = __builtin_fmaf(wl[k * k + u * k + v], p1[(long)rr * S + ss], acc);
        }
    }
    out[(long)b * RS + idx] = sigmoidf_(fmaxf(acc * aff[0] + aff[1], 0.f));
}

// ---- BAM ---------------------------------------------------------------------------------------------------------------------
// channel gate: BatchNorm1d(eval, folded)( W2 relu(W1 mean + b1) + b2 ), one workgroup per image
__global__ __launch_bounds__(256) void bam_channel_kernel(const float* __restrict__ mean, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ bn_s,
                                                         const float* __restrict__ bn_t, float* __restrict__ cg, int C, int Cr) {
    extern __shared__ float hbuf[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* mv = mean + (long)b * C;
    for (int k = wave; k < Cr; k += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = __builtin_fmaf(w1[(long)k * C + c], mv[c], s);
        s = wave_sum(s);
        if (lane == 0) hbuf[k] = fmaxf(s + b1[k], 0.f);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float s = b2[c];
        for (int k = 0; k < Cr; ++k) s = __builtin_fmaf(w2[(long)c * Cr + k], hbuf[k], s);
        cg[(long)b * C + c] = s * bn_s[c] + bn_t[c];
    }
}

// dilated 3x3 convolution Cr -> Cr (zero padding = dilation) + folded affine + ReLU on (B, Cr, H, W).  One thread computes all
// CMAX >= Cr output channels of four consecutive pixels, so that every (wave-uniform, scalar-loaded) weight feeds four FMAs.
template <int CMAX>
__global__ __launch_bounds__(256) void small_conv_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, float* __restrict__ out, int Cr, int H, int W, int dil) {
    constexpr int PIX = 4;
    const long HW = (long)H * W;
    const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * PIX;
    if (p0 >= HW) return;
    const int b = blockIdx.y;
    int pi[PIX], pj[PIX];
#pragma unroll
    for (int e = 0; e < PIX; ++e) {
        const long p = p0 + e < HW ? p0 + e : HW - 1;
        pi[e] = (int)(p / W); pj[e] = (int)(p - (long)pi[e] * W);
    }
    const float* ip = in + (long)b * Cr * HW;
    float acc[CMAX][PIX];
#pragma unroll
    for (int o = 0; o < CMAX; ++o)
#pragma unroll
        for (int e = 0; e < PIX; ++e) acc[o][e] = 0.f;
    for (int ci = 0; ci < Cr; ++ci) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                float xv[PIX];
#pragma unroll
                for (int e = 0; e < PIX; ++e) {
                    const int ii = pi[e] + (u - 1) * dil, jj = pj[e] + (v - 1) * dil;
                    xv[e] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? ip[(long)ci * HW + (long)ii * W + jj] : 0.f;
                }
#pragma unroll
                for (int o = 0; o < CMAX; ++o)
                    if (o < Cr) {
                        const float wk = w[(((long)o * Cr + ci) * 3 + u) * 3 + v];
#pragma unroll
                        for (int e = 0; e < PIX; ++e) acc[o][e] = __builtin_fmaf(wk, xv[e], acc[o][e]);
                    }
            }
        }
    }
    float* op = out + (long)b * Cr * HW + p0;
#pragma unroll
    for (int o = 0; o < CMAX; ++o)
        if (o < Cr) {
#pragma unroll
            for (int e = 0; e < PIX; ++e)
                if (p0 + e < HW) op[(long)o * HW + e] = fmaxf(acc[o][e] * sc[o] + sh[o], 0.f);
        }
}

size_t fl(size_t n) { return (n + 63) & ~(size_t)63; }           // 256-byte aligned sub-buffers (in floats)

}  // namespace

extern "C" {

size_t mi355_axis_attn_workspace_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const size_t bc = (size_t)B * C, hw = (size_t)H * W;
    // four (B,C,H) + four (B,C,W) axis maps, three (B,HW) planes, three (B,C) vectors
    return 4 * (4 * fl(bc * H) + 4 * fl(bc * W) + 3 * fl((size_t)B * hw) + 3 * fl(bc)) + 256;
}

int mi355_gc_fwd(const float* x, const float* conv_w, const float* conv_b, const float* w1, const float* b1, const float* ln_w,
                 const float* ln_b, const float* w2, const float* b2, float* y, int B, int C, int Cr, int H, int W, float ln_eps,
                 void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && conv_w && w1 && ln_w && ln_b && w2 && y && workspace && B > 0 && C > 0 && Cr > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(workspace_bytes >= mi355_axis_attn_workspace_bytes(B, C, H, W) && aligned16(workspace));
    if (Cr > 8192) return mi355::fail(MI355_EUNSUPPORTED, "mi355_gc_fwd: hidden width %d > 8192", Cr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* attn = ws;                                   // (B, HW)   conv(x): gc_module.py:33
    float* ctx = attn + fl((size_t)B * HW);             // (B, C)    matmul(input_x, context): :35
    float* tvec = ctx + fl((size_t)B * C);              // (B, C)    transform(context): :41
    launch_chan_reduce<0, 1>(x, conv_w, conv_b, attn, B, C, HW, 1, st);
    launch_plane_dot(x, attn, ctx, (long)B * C, C, HW, 1.0f, st);
    gc_transform_kernel<<<B, 256, Cr * sizeof(float), st>>>(ctx, w1, b1, ln_w, ln_b, w2, b2, tvec, C, Cr, ln_eps);
    ApplyArgs g{};
    g.x = x; g.y = y; g.a = tvec; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_GC>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_coordatt_fwd(const float* x, const float* w1, const float* b1, const float* bn_scale, const float* bn_shift, const float* wh,
                       const float* bh, const float* ww, const float* bw, float* y, int B, int C, int hidden, int H, int W,
                       void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w1 && bn_scale && bn_shift && wh && ww && y && workspace && B > 0 && C > 0 && hidden > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(workspace_bytes >= mi355_axis_attn_workspace_bytes(B, C, H, W) && aligned16(workspace));
    if (hidden > 128) return mi355::fail(MI355_EUNSUPPORTED, "mi355_coordatt_fwd: hidden width %d > 128", hidden);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bc = (size_t)B * C;
    float* ws = static_cast<float*>(workspace);
    float* ph = ws;                       // (B,C,H)  pool_h: coordatten.py:33
    float* pw = ph + fl(bc * H);          // (B,C,W)  pool_w: :34
    float* ah = pw + fl(bc * W);          // (B,C,H)  conv_h(...): :41
    float* aw = ah + fl(bc * H);          // (B,C,W)  conv_w(...): :42
    const int rc = launch_plane_pool<false>(x, ph, nullptr, pw, nullptr, (long)bc, H, W, st);
    if (rc != MI355_OK) return rc;
    coord_mlp_kernel<<<dim3(cdiv(H + W, 64), B), 256, (size_t)hidden * 64 * sizeof(float), st>>>(ph, pw, w1, b1, bn_scale, bn_shift, wh, bh, ww, bw,
                                                                                                 ah, aw, C, hidden, H, W);
    ApplyArgs g{};
    g.x = x; g.y = y; g.a = ah; g.b = aw; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_COORD>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

int mi355_triplet_fwd(const float* x, const float* w_ch, const float* w_cw, const float* w_hw, const float* affine, float* y, int B, int C,
                      int H, int W, int ksize, void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w_ch && w_cw && w_hw && affine && y && workspace && B > 0 && C > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(ksize >= 1 && ksize <= 15 && (ksize & 1));
    MI355_CHECK_ARG(workspace_bytes >= mi355_axis_attn_workspace_bytes(B, C, H, W) && aligned16(workspace));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bc = (size_t)B * C;
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* h_mean = ws;                            // ZPool over w of x.permute(0,3,1,2): (B,C,H) mean / max   (triplet_attention.py:59, :33-36)
    float* h_max = h_mean + fl(bc * H);
    float* w_mean = h_max + fl(bc * H);            // ZPool over h of x.permute(0,2,1,3): (B,C,W)              (:60)
    float* w_max = w_mean + fl(bc * W);
    float* s_ch = w_max + fl(bc * W);              // gates
    float* s_cw = s_ch + fl(bc * H);
    float* zp = s_cw + fl(bc * W);                 // ZPool over c: (B,2,HW)                                   (:61)
    float* s_hw = zp + 2 * fl((size_t)B * HW);
    const int rc = launch_plane_pool<true>(x, h_mean, h_max, w_mean, w_max, (long)bc, H, W, st);
    if (rc != MI355_OK) return rc;
    launch_chan_reduce<1, 1>(x, nullptr, nullptr, zp, B, C, HW, 2, st);
    gate_conv_kernel<<<dim3(cdiv((long)C * H, 256), B), 256, 0, st>>>(h_mean, h_max, (long)C * H, w_ch, affine + 0, s_ch, C, H, ksize);
    gate_conv_kernel<<<dim3(cdiv((long)C * W, 256), B), 256, 0, st>>>(w_mean, w_max, (long)C * W, w_cw, affine + 2, s_cw, C, W, ksize);
    gate_conv_kernel<<<dim3(cdiv(HW, 256), B), 256, 0, st>>>(zp, zp + HW, 2 * HW, w_hw, affine + 4, s_hw, H, W, ksize);
    ApplyArgs g{};
    g.x = x; g.y = y; g.a = s_ch; g.b = s_cw; g.c = s_hw; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_TRIPLET>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

size_t mi355_bam_workspace_bytes(int B, int C, int Cr, int H, int W) {
    if (B <= 0 || C <= 0 || Cr <= 0 || H <= 0 || W <= 0) return 0;
    const size_t hw = (size_t)H * W;
    return 4 * (2 * fl((size_t)B * C) + 2 * fl((size_t)B * Cr * hw) + fl((size_t)B * hw)) + 256;
}

int mi355_bam_fwd(const float* x, const float* const* p, float* y, int B, int C, int Cr, int H, int W, int dilation, void* workspace,
                  size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && p && y && workspace && B > 0 && C > 0 && Cr > 0 && H > 0 && W > 0 && dilation > 0);
    for (int q = 0; q < MI355_BAM_NPARAMS; ++q) MI355_CHECK_ARG(p[q] != nullptr);
    MI355_CHECK_ARG(workspace_bytes >= mi355_bam_workspace_bytes(B, C, Cr, H, W) && aligned16(workspace));
    if (Cr > 32) return mi355::fail(MI355_EUNSUPPORTED, "mi355_bam_fwd: reduced width %d > 32", Cr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* mean = ws;                                    // (B,C)      avgpool: bam.py:30
    float* cg = mean + fl((size_t)B * C);                // (B,C)      bn(mlp(.)): :31-32
    float* t0 = cg + fl((size_t)B * C);                  // (B,Cr,HW)  conv1: :55
    float* t1 = t0 + fl((size_t)B * Cr * HW);            // (B,Cr,HW)  conv2 stages: :56
    float* sg = t1 + fl((size_t)B * Cr * HW);            // (B,HW)     bn(conv3(.)): :57-58
    launch_plane_dot(x, nullptr, mean, (long)B * C, C, HW, 1.0f / (float)HW, st);
    bam_channel_kernel<<<B, 256, Cr * sizeof(float), st>>>(mean, p[MI355_BAM_FC1_W], p[MI355_BAM_FC1_B], p[MI355_BAM_FC2_W], p[MI355_BAM_FC2_B],
                                                           p[MI355_BAM_BN1D_SCALE], p[MI355_BAM_BN1D_SHIFT], cg, C, Cr);
#define CR_DISPATCH(FN)                                  \
    do {                                                 \
        if (Cr <= 4) FN(4); else if (Cr <= 8) FN(8); else if (Cr <= 16) FN(16); else FN(32); \
    } while (0)
#define RED(K_) launch_chan_reduce<0, K_>(x, p[MI355_BAM_CONV1_W], p[MI355_BAM_CONV1_B], t0, B, C, HW, Cr, st)
    CR_DISPATCH(RED);
#undef RED
    const dim3 grid(cdiv(cdiv(HW, 4), 256), B);
#define SC1(K_) small_conv_kernel<K_><<<grid, 256, 0, st>>>(t0, p[MI355_BAM_DCONV1_W], p[MI355_BAM_DCONV1_SCALE], p[MI355_BAM_DCONV1_SHIFT], t1, Cr, H, W, dilation)
    CR_DISPATCH(SC1);
#undef SC1
#define SC2(K_) small_conv_kernel<K_><<<grid, 256, 0, st>>>(t1, p[MI355_BAM_DCONV2_W], p[MI355_BAM_DCONV2_SCALE], p[MI355_BAM_DCONV2_SHIFT], t0, Cr, H, W, dilation)
    CR_DISPATCH(SC2);
#undef SC2
#undef CR_DISPATCH
    launch_chan_reduce<0, 1>(t0, p[MI355_BAM_CONV3_W], p[MI355_BAM_CONV3_B], sg, B, Cr, HW, 1, st);
    ApplyArgs g{};
    g.x = x; g.y = y; g.a = cg; g.b = sg; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_BAM>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
