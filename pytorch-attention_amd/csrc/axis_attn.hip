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

enum { AP_GC = 0, AP_COORD = 1, AP_TRIPLET = 2, AP_BAM = 3, AP_SPATIAL = 4 };

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
    const long p_raw = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    const bool live = p_raw < HW;
    const long p = live ? p_raw : 0;                             // idle threads shadow pixel 0 (they take part in the LDS staging barrier)
    const int b = blockIdx.y;
    const float* xp = x + (long)b * C * HW + p;
    constexpr int NA = MODE == 0 ? KMAX : 2;
    constexpr int U = (MODE == 0 && KMAX >= 16) ? 4 : 8;
    V acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = splat(V{}, (MODE == 1 && k == 1) ? -INFINITY : 0.f);
    // wide outputs: the (K, C) weight is staged transposed in LDS ([c][KMAX], zero rows past K) so that a channel's KMAX weights are
    // a few broadcast 16-byte reads instead of KMAX strided scalar loads
    extern __shared__ __attribute__((aligned(16))) float wl[];
    if constexpr (MODE == 0 && KMAX >= 4) {
        for (int q = threadIdx.x; q < C * KMAX; q += 256) {
            const int c = q / KMAX, k = q - c * KMAX;
            wl[q] = k < K ? w[(long)k * C + c] : 0.f;
        }
        __syncthreads();
    }
    auto sweep = [&](int cbeg, int cend) {                       // channels [cbeg, cend) into acc, U loads in flight
        for (int c = cbeg; c < cend; c += U) {
            V v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const V*>(xp + (long)(c + u < cend ? c + u : cend - 1) * HW);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = c + u < cend;
                if constexpr (MODE == 0 && KMAX >= 4) {
                    if (in) {
                        const f4* wr = reinterpret_cast<const f4*>(wl + (c + u) * KMAX);
#pragma unroll
                        for (int k4 = 0; k4 < KMAX / 4; ++k4) {
                            const f4 wk = wr[k4];
                            acc[k4 * 4 + 0] += wk.x * v[u]; acc[k4 * 4 + 1] += wk.y * v[u];
                            acc[k4 * 4 + 2] += wk.z * v[u]; acc[k4 * 4 + 3] += wk.w * v[u];
                        }
                    }
                } else if constexpr (MODE == 0) {
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
    };
    if constexpr (KMAX == 1) {
        // Every image of a batch has the same layout, and C*HW*4 bytes is often a multiple of the HBM channel-interleave period
        // (256 x 56 x 56 x 4 = 49 x 64 KB), so workgroups sweeping the channels in lock step would all queue on the same few
        // memory channels.  The channel axis is cut into eight groups and image b starts at group b mod 8; each group is summed in
        // its own fixed order and the eight partials are combined in a fixed order, so the result does not depend on b.
        constexpr int G = 8;
        const int gs = ((C + G - 1) / G + U - 1) / U * U;
        V part[G][NA];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < NA; ++k) part[g][k] = splat(V{}, (MODE == 1 && k == 1) ? -INFINITY : 0.f);
        for (int step = 0; step < G; ++step) {
            const int g = (b + step) & (G - 1);
            const int cbeg = g * gs, cend = cbeg + gs < C ? cbeg + gs : C;
#pragma unroll
            for (int k = 0; k < NA; ++k) acc[k] = splat(V{}, (MODE == 1 && k == 1) ? -INFINITY : 0.f);
            if (cbeg < cend) sweep(cbeg, cend);
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int k = 0; k < NA; ++k) part[q][k] = q == g ? acc[k] : part[q][k];
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if (MODE == 1 && k == 1)
                acc[k] = vmaxf(vmaxf(vmaxf(part[0][k], part[1][k]), vmaxf(part[2][k], part[3][k])),
                               vmaxf(vmaxf(part[4][k], part[5][k]), vmaxf(part[6][k], part[7][k])));
            else
                acc[k] = ((part[0][k] + part[1][k]) + (part[2][k] + part[3][k])) + ((part[4][k] + part[5][k]) + (part[6][k] + part[7][k]));
        }
    } else {
        sweep(0, C);
    }
    const int nout = MODE == 0 ? K : 2;
    float* op = out + (long)b * nout * HW + p;
    if (!live) return;
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
    const size_t lds = (MODE == 0 && KMAX >= 4) ? (size_t)C * KMAX * sizeof(float) : 0;
    if (vec) chan_reduce_kernel<MODE, KMAX, 4><<<dim3(cdiv(HW / 4, 256), B), 256, lds, st>>>(x, w, bias, out, C, HW, K);
    else     chan_reduce_kernel<MODE, KMAX, 1><<<dim3(cdiv(HW, 256), B), 256, lds, st>>>(x, w, bias, out, C, HW, K);
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

// The same reductions for planes that fit LDS (H * (W + 1) floats <= 64 KB): one workgroup per plane streams it in with 16-byte
// loads (a few in flight per thread), parks it in LDS with an odd pitch, then half of the threads reduce rows and the other half
// columns, both conflict-free.
template <bool WITH_MAX, int VEC>
__global__ __launch_bounds__(256) void plane_pool_lds_kernel(const float* __restrict__ x, float* __restrict__ h_mean, float* __restrict__ h_max,
                                                            float* __restrict__ w_mean, float* __restrict__ w_max, int H, int W) {
    extern __shared__ float tile[];
    const long plane = blockIdx.x;
    const int t = threadIdx.x, pitch = W + 1;
    const float* xp = x + plane * (long)H * W;
    if constexpr (VEC == 4) {
        const int w4 = W >> 2, n4 = H * w4;
        int row = t / w4, col = t - row * w4;
        const int drow = 256 / w4, dcol = 256 - drow * w4;
        for (int i = t; i < n4; i += 256) {
            const f4 v = *reinterpret_cast<const f4*>(xp + (long)i * 4);
            float* d = tile + row * pitch + col * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            row += drow; col += dcol;
            if (col >= w4) { col -= w4; ++row; }
        }
    } else {
        const int n = H * W;
        int row = t / W, col = t - row * W;
        const int drow = 256 / W, dcol = 256 - drow * W;
        for (int i = t; i < n; i += 256) {
            tile[row * pitch + col] = xp[i];
            row += drow; col += dcol;
            if (col >= W) { col -= W; ++row; }
        }
    }
    __syncthreads();
    if (t < 128) {
        for (int r = t; r < H; r += 128) {
            const float* q = tile + r * pitch;
            float s = 0.f, m = -INFINITY;
            for (int j = 0; j < W; ++j) { s += q[j]; if constexpr (WITH_MAX) m = fmaxf(m, q[j]); }
            h_mean[plane * H + r] = s / (float)W;
            if constexpr (WITH_MAX) h_max[plane * H + r] = m;
        }
    } else {
        for (int j = t - 128; j < W; j += 128) {
            const float* q = tile + j;
            float s = 0.f, m = -INFINITY;
            for (int r = 0; r < H; ++r) { s += q[r * pitch]; if constexpr (WITH_MAX) m = fmaxf(m, q[r * pitch]); }
            w_mean[plane * W + j] = s / (float)H;
            if constexpr (WITH_MAX) w_max[plane * W + j] = m;
        }
    }
}

template <bool WITH_MAX>
int launch_plane_pool(const float* x, float* h_mean, float* h_max, float* w_mean, float* w_max, long planes, int H, int W, hipStream_t st) {
    const size_t lds = (size_t)H * (W + 1) * sizeof(float);
    if (lds <= 65536 && planes < (1L << 31)) {
        if ((W & 3) == 0 && aligned16(x)) plane_pool_lds_kernel<WITH_MAX, 4><<<(int)planes, 256, lds, st>>>(x, h_mean, h_max, w_mean, w_max, H, W);
        else                              plane_pool_lds_kernel<WITH_MAX, 1><<<(int)planes, 256, lds, st>>>(x, h_mean, h_max, w_mean, w_max, H, W);
        return MI355_OK;
    }
    const int grid = cdiv(planes, 4);
    if (W <= 64)       plane_pool_kernel<WITH_MAX, 1><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else if (W <= 128) plane_pool_kernel<WITH_MAX, 2><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else if (W <= 256) plane_pool_kernel<WITH_MAX, 4><<<grid, 256, 0, st>>>(x, h_mean, h_max, w_mean, w_max, planes, H, W);
    else return mi355::fail(MI355_EUNSUPPORTED, "axis pooling: W = %d > 256 with a plane larger than 64 KB", W);
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
        for (long q0 = lane; q0 < n4; q0 += 256) {               // four 16-byte loads of x (and of v) in flight per lane
            f4 a[4], u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long q = q0 + 64 * k < n4 ? q0 + 64 * k : q0;
                a[k] = *reinterpret_cast<const f4*>(xp + q * 4);
                if (vp) u[k] = *reinterpret_cast<const f4*>(vp + q * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (q0 + 64 * k < n4) {
                    const float d = vp ? a[k].x * u[k].x + a[k].y * u[k].y + a[k].z * u[k].z + a[k].w * u[k].w
                                       : (a[k].x + a[k].y) + (a[k].z + a[k].w);
                    if (k & 1) s1 += d; else s0 += d;
                }
            }
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
    } else if constexpr (MODE == AP_SPATIAL) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) yv[e] = xv[e] * g.c[img * HW + p + e];
    } else {
        const float cg = g.a[plane];
#pragma unroll
        for (int e = 0; e < VEC; ++e) yv[e] = xv[e] + xv[e] * sigmoidf_(cg + g.b[img * HW + p + e]);
    }
    float* yp = g.y + plane * HW + p;
    if constexpr (VEC == 4) *reinterpret_cast<f4*>(yp) = f4{yv[0], yv[1], yv[2], yv[3]};
    else yp[0] = yv[0];
}

// One element group per thread and nothing else: measured 6.4 TB/s of combined read + write at the C2 shape, above the float4 copy
// yardstick; a wave-per-plane variant with four loads in flight per lane and no integer divisions was 12 % slower.
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
        if (valid) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int c = 0;
            for (; c + 8 <= C; c += 8) {                         // eight independent loads in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(long)(c + u) * ld];
                const float* wr = w1 + (long)k * C + c;
                s = __builtin_fmaf(wr[0], v[0], s);   s1 = __builtin_fmaf(wr[1], v[1], s1);
                s2 = __builtin_fmaf(wr[2], v[2], s2); s3 = __builtin_fmaf(wr[3], v[3], s3);
                s = __builtin_fmaf(wr[4], v[4], s);   s1 = __builtin_fmaf(wr[5], v[5], s1);
                s2 = __builtin_fmaf(wr[6], v[6], s2); s3 = __builtin_fmaf(wr[7], v[7], s3);
            }
            for (; c < C; ++c) s = __builtin_fmaf(w1[(long)k * C + c], src[(long)c * ld], s);
            s = (s + s1) + (s2 + s3);
        }
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
template <int KS>
__global__ __launch_bounds__(256) void gate_conv_kernel(const float* __restrict__ in0, const float* __restrict__ in1, long bs, const float* __restrict__ w,
                                                       const float* __restrict__ aff, float* __restrict__ out, int R, int S, int kdyn) {
    // a thread computes four consecutive outputs of one row: per input row it loads the k + 3 values under the sliding window once
    constexpr int KMAXW = KS > 0 ? KS : 15;
    const int k = KS > 0 ? KS : kdyn;
    __shared__ float wl[2 * 15 * 15];
    for (int q = threadIdx.x; q < 2 * k * k; q += 256) wl[q] = w[q];
    __syncthreads();
    const int s4 = (S + 3) >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)R * s4) return;
    const int b = blockIdx.y, r = (int)(idx / s4), s0 = (int)(idx - (long)r * s4) * 4, pad = (k - 1) / 2;
    const float* p0 = in0 + (long)b * bs;
    const float* p1 = in1 + (long)b * bs;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < k; ++u) {
        const int rr = r + u - pad;
        if (rr < 0 || rr >= R) continue;
        float a[KMAXW + 3], m[KMAXW + 3];
#pragma unroll
        for (int q = 0; q < KMAXW + 3; ++q) {
            const int ss = s0 + q - pad;
            const bool in = q < k + 3 && ss >= 0 && ss < S;
            a[q] = in ? p0[(long)rr * S + ss] : 0.f;
            m[q] = in ? p1[(long)rr * S + ss] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < KMAXW; ++v) {
            if (v < k) {
                const float w0 = wl[u * k + v], w1 = wl[k * k + u * k + v];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(w1, m[v + e], __builtin_fmaf(w0, a[v + e], acc[e]));
            }
        }
    }
    float* op = out + (long)b * R * S + (long)r * S + s0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (s0 + e < S) op[e] = sigmoidf_(fmaxf(acc[e] * aff[0] + aff[1], 0.f));
}

// The same gate with the input tile staged in LDS: a workgroup owns TR rows x all S columns of one image's map; both planes are parked
// with a zero halo (KS - 1 rows, 4 columns to the left so that a thread's 12-float window starts 16-byte aligned); a thread computes four
// consecutive outputs of a row from 3 aligned 16-byte reads per plane and input row.
template <int KS>
__global__ __launch_bounds__(256) void gate_conv_lds_kernel(const float* __restrict__ in0, const float* __restrict__ in1, long bs, const float* __restrict__ w,
                                                           const float* __restrict__ aff, float* __restrict__ out, int R, int S, int TR, int PW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PAD = (KS - 1) / 2, OFF = 4 - PAD;
    const int rows = TR + KS - 1, plane = rows * PW;
    float* wl = lds + 2 * plane;                                  // 2 * KS * KS weights
    const int t = threadIdx.x, b = blockIdx.y, r0 = blockIdx.x * TR;
    for (int q = t; q < 2 * plane; q += 256) lds[q] = 0.f;
    for (int q = t; q < 2 * KS * KS; q += 256) wl[q] = w[q];
    __syncthreads();
    const float* p0 = in0 + (long)b * bs;
    const float* p1 = in1 + (long)b * bs;
    for (int q = t; q < rows * S; q += 256) {
        const int rr = q / S, col = q - rr * S, r = r0 - PAD + rr;
        if (r >= 0 && r < R) {
            lds[rr * PW + 4 + col] = p0[(long)r * S + col];
            lds[plane + rr * PW + 4 + col] = p1[(long)r * S + col];
        }
    }
    __syncthreads();
    const int s4 = (S + 3) >> 2;
    const int tr = t / s4, q4 = t - tr * s4;
    if (tr >= TR || r0 + tr >= R) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u = 0; u < KS; ++u) {
        const float* ra = lds + (tr + u) * PW + q4 * 4;
        const float* rm = ra + plane;
        float a[12], m[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f4 va = *reinterpret_cast<const f4*>(ra + c * 4), vm = *reinterpret_cast<const f4*>(rm + c * 4);
            a[c * 4] = va.x; a[c * 4 + 1] = va.y; a[c * 4 + 2] = va.z; a[c * 4 + 3] = va.w;
            m[c * 4] = vm.x; m[c * 4 + 1] = vm.y; m[c * 4 + 2] = vm.z; m[c * 4 + 3] = vm.w;
        }
#pragma unroll
        for (int v = 0; v < KS; ++v) {
            const float w0 = wl[u * KS + v], w1 = wl[KS * KS + u * KS + v];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(w1, m[OFF + e + v], __builtin_fmaf(w0, a[OFF + e + v], acc[e]));
        }
    }
    float* op = out + (long)b * R * S + (long)(r0 + tr) * S + q4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q4 * 4 + e < S) op[e] = sigmoidf_(fmaxf(acc[e] * aff[0] + aff[1], 0.f));
}

template <int KS>
bool try_gate_conv_lds(const float* in0, const float* in1, long bs, const float* w, const float* aff, float* out, int B, int R, int S, hipStream_t st) {
    const int s4 = (S + 3) >> 2;
    if (s4 > 256) return false;
    int TR = 256 / s4;
    if (TR > R) TR = R;
    const int PW = s4 * 4 + 8;                                    // 4 columns of halo on the left, >= 4 + PAD readable on the right
    const size_t lds = ((size_t)2 * (TR + KS - 1) * PW + 2 * KS * KS) * sizeof(float);
    if (lds > 65536) return false;
    gate_conv_lds_kernel<KS><<<dim3(cdiv(R, TR), B), 256, lds, st>>>(in0, in1, bs, w, aff, out, R, S, TR, PW);
    return true;
}

void launch_gate_conv(const float* in0, const float* in1, long bs, const float* w, const float* aff, float* out, int B, int R, int S, int k,
                      hipStream_t st) {
    if (k == 7 && try_gate_conv_lds<7>(in0, in1, bs, w, aff, out, B, R, S, st)) return;
    if (k == 5 && try_gate_conv_lds<5>(in0, in1, bs, w, aff, out, B, R, S, st)) return;
    if (k == 3 && try_gate_conv_lds<3>(in0, in1, bs, w, aff, out, B, R, S, st)) return;
    const dim3 grid(cdiv((long)R * ((S + 3) >> 2), 256), B);
    if (k == 7)      gate_conv_kernel<7><<<grid, 256, 0, st>>>(in0, in1, bs, w, aff, out, R, S, k);
    else if (k == 3) gate_conv_kernel<3><<<grid, 256, 0, st>>>(in0, in1, bs, w, aff, out, R, S, k);
    else             gate_conv_kernel<0><<<grid, 256, 0, st>>>(in0, in1, bs, w, aff, out, R, S, k);
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
// CMAX >= Cr output channels of four (two for CMAX = 32) consecutive pixels; the weights sit in LDS as [ci][tap][CMAX] and come in
// as broadcast 16-byte reads.
template <int CMAX>
__global__ __launch_bounds__(256) void small_conv_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, float* __restrict__ out, int Cr, int H, int W, int dil) {
    constexpr int PIX = CMAX >= 32 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [ci][3][3][CMAX], zero past Cr
    for (int q = threadIdx.x; q < Cr * 9 * CMAX; q += 256) {
        const int o = q % CMAX, tap = (q / CMAX) % 9, ci = q / (CMAX * 9);
        wl[q] = o < Cr ? w[((long)o * Cr + ci) * 9 + tap] : 0.f;
    }
    __syncthreads();
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
                const f4* wr = reinterpret_cast<const f4*>(wl + ((ci * 3 + u) * 3 + v) * CMAX);
#pragma unroll
                for (int o4 = 0; o4 < CMAX / 4; ++o4) {
                    const f4 wk = wr[o4];
#pragma unroll
                    for (int e = 0; e < PIX; ++e) {
                        acc[o4 * 4 + 0][e] = __builtin_fmaf(wk.x, xv[e], acc[o4 * 4 + 0][e]);
                        acc[o4 * 4 + 1][e] = __builtin_fmaf(wk.y, xv[e], acc[o4 * 4 + 1][e]);
                        acc[o4 * 4 + 2][e] = __builtin_fmaf(wk.z, xv[e], acc[o4 * 4 + 2][e]);
                        acc[o4 * 4 + 3][e] = __builtin_fmaf(wk.w, xv[e], acc[o4 * 4 + 3][e]);
                    }
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

// The same convolution when W % 4 == 0 and dilation % 4 == 0 (BAM's default: dilation 4): the three taps of a row sit exactly one
// aligned pixel quad to the left, at, and to the right of the thread's own quad, so a tap is ONE 16-byte load that is either
// fully inside the image or fully outside -- 9 loads per input channel instead of 36 bounds-checked scalars.
template <int CMAX>
__global__ __launch_bounds__(256) void small_conv_quad_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ sc,
                                                             const float* __restrict__ sh, float* __restrict__ out, int Cr, int H, int W, int dil) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [ci][3][3][CMAX], zero past Cr
    for (int q = threadIdx.x; q < Cr * 9 * CMAX; q += 256) {
        const int o = q % CMAX, tap = (q / CMAX) % 9, ci = q / (CMAX * 9);
        wl[q] = o < Cr ? w[((long)o * Cr + ci) * 9 + tap] : 0.f;
    }
    __syncthreads();
    const long HW = (long)H * W;
    const int w4 = W >> 2;
    const long qid = (long)blockIdx.x * 256 + threadIdx.x;
    if (qid >= (long)H * w4) return;
    const int b = blockIdx.y, i = (int)(qid / w4), j = (int)(qid - (long)i * w4) * 4;
    const float* ip = in + (long)b * Cr * HW;
    f4 acc[CMAX];
#pragma unroll
    for (int o = 0; o < CMAX; ++o) acc[o] = f4{0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < Cr; ++ci) {
        f4 xv[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ii = i + (u - 1) * dil;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int jj = j + (v - 1) * dil;
                const bool in_img = ii >= 0 && ii < H && jj >= 0 && jj < W;
                xv[u * 3 + v] = *reinterpret_cast<const f4*>(ip + (long)ci * HW + (long)(in_img ? ii : i) * W + (in_img ? jj : j));
                if (!in_img) xv[u * 3 + v] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f4* wr = reinterpret_cast<const f4*>(wl + (ci * 9 + tap) * CMAX);
#pragma unroll
            for (int o4 = 0; o4 < CMAX / 4; ++o4) {
                const f4 wk = wr[o4];
                acc[o4 * 4 + 0] += wk.x * xv[tap]; acc[o4 * 4 + 1] += wk.y * xv[tap];
                acc[o4 * 4 + 2] += wk.z * xv[tap]; acc[o4 * 4 + 3] += wk.w * xv[tap];
            }
        }
    }
    float* op = out + (long)b * Cr * HW + (long)i * W + j;
#pragma unroll
    for (int o = 0; o < CMAX; ++o)
        if (o < Cr) {
            const f4 r = acc[o] * sc[o] + sh[o];
            *reinterpret_cast<f4*>(op + (long)o * HW) = f4{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
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
    launch_gate_conv(h_mean, h_max, (long)C * H, w_ch, affine + 0, s_ch, B, C, H, ksize, st);
    launch_gate_conv(w_mean, w_max, (long)C * W, w_cw, affine + 2, s_cw, B, C, W, ksize, st);
    launch_gate_conv(zp, zp + HW, 2 * HW, w_hw, affine + 4, s_hw, B, H, W, ksize, st);
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

// The two gates of BAM (ChannelGate.forward bam.py:28-33, SpatialGate.forward :53-59) without the broadcast: cg (B,C) and / or
// sg (B,HW); either output may be null.  t0 / t1: (B,Cr,HW) scratch, mean: (B,C) scratch.
static int bam_gates(const float* x, const float* const* p, float* mean, float* cg, float* t0, float* t1, float* sg, int B, int C, int Cr,
                     int H, int W, int dilation, hipStream_t st) {
    const long HW = (long)H * W;
    if (cg) {
        launch_plane_dot(x, nullptr, mean, (long)B * C, C, HW, 1.0f / (float)HW, st);
        bam_channel_kernel<<<B, 256, Cr * sizeof(float), st>>>(mean, p[MI355_BAM_FC1_W], p[MI355_BAM_FC1_B], p[MI355_BAM_FC2_W], p[MI355_BAM_FC2_B],
                                                               p[MI355_BAM_BN1D_SCALE], p[MI355_BAM_BN1D_SHIFT], cg, C, Cr);
    }
    if (!sg) return MI355_OK;
#define CR_DISPATCH(FN)                                  \
    do {                                                 \
        if (Cr <= 4) FN(4); else if (Cr <= 8) FN(8); else if (Cr <= 16) FN(16); else FN(32); \
    } while (0)
#define RED(K_) launch_chan_reduce<0, K_>(x, p[MI355_BAM_CONV1_W], p[MI355_BAM_CONV1_B], t0, B, C, HW, Cr, st)
    CR_DISPATCH(RED);
#undef RED
    const dim3 grid(cdiv(cdiv(HW, Cr > 16 ? 2 : 4), 256), B);
    const bool quad = (W & 3) == 0 && (dilation & 3) == 0 && Cr <= 16;
#define SCQ(IN_, W_, SC_, SH_, OUT_)                                                                                                \
    do {                                                                                                                           \
        const dim3 qgrid(cdiv((long)H * (W >> 2), 256), B);                                                                        \
        if (Cr <= 4)      small_conv_quad_kernel<4><<<qgrid, 256, (size_t)Cr * 9 * 4 * sizeof(float), st>>>(IN_, W_, SC_, SH_, OUT_, Cr, H, W, dilation);  \
        else if (Cr <= 8) small_conv_quad_kernel<8><<<qgrid, 256, (size_t)Cr * 9 * 8 * sizeof(float), st>>>(IN_, W_, SC_, SH_, OUT_, Cr, H, W, dilation);  \
        else              small_conv_quad_kernel<16><<<qgrid, 256, (size_t)Cr * 9 * 16 * sizeof(float), st>>>(IN_, W_, SC_, SH_, OUT_, Cr, H, W, dilation); \
    } while (0)
#define SC1(K_) small_conv_kernel<K_><<<grid, 256, (size_t)Cr * 9 * K_ * sizeof(float), st>>>(t0, p[MI355_BAM_DCONV1_W], p[MI355_BAM_DCONV1_SCALE], p[MI355_BAM_DCONV1_SHIFT], t1, Cr, H, W, dilation)
    if (quad) SCQ(t0, p[MI355_BAM_DCONV1_W], p[MI355_BAM_DCONV1_SCALE], p[MI355_BAM_DCONV1_SHIFT], t1);
    else      CR_DISPATCH(SC1);
#undef SC1
#define SC2(K_) small_conv_kernel<K_><<<grid, 256, (size_t)Cr * 9 * K_ * sizeof(float), st>>>(t1, p[MI355_BAM_DCONV2_W], p[MI355_BAM_DCONV2_SCALE], p[MI355_BAM_DCONV2_SHIFT], t0, Cr, H, W, dilation)
    if (quad) SCQ(t1, p[MI355_BAM_DCONV2_W], p[MI355_BAM_DCONV2_SCALE], p[MI355_BAM_DCONV2_SHIFT], t0);
    else      CR_DISPATCH(SC2);
#undef SC2
#undef SCQ
#undef CR_DISPATCH
    launch_chan_reduce<0, 1>(t0, p[MI355_BAM_CONV3_W], p[MI355_BAM_CONV3_B], sg, B, Cr, HW, 1, st);
    return MI355_OK;
}

static int bam_check(const float* x, const float* const* p, int B, int C, int Cr, int H, int W, int dilation, const void* workspace,
                     size_t workspace_bytes) {
    MI355_CHECK_ARG(x && p && workspace && B > 0 && C > 0 && Cr > 0 && H > 0 && W > 0 && dilation > 0);
    for (int q = 0; q < MI355_BAM_NPARAMS; ++q) MI355_CHECK_ARG(p[q] != nullptr);
    MI355_CHECK_ARG(workspace_bytes >= mi355_bam_workspace_bytes(B, C, Cr, H, W) && aligned16(workspace));
    if (Cr > 32) return mi355::fail(MI355_EUNSUPPORTED, "mi355_bam_fwd: reduced width %d > 32", Cr);
    const int kmax = Cr <= 4 ? 4 : (Cr <= 8 ? 8 : (Cr <= 16 ? 16 : 32));
    if ((size_t)C * kmax * sizeof(float) > 65536)
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_bam_fwd: C = %d with reduced width %d exceeds the 64 KB weight stage", C, Cr);
    return MI355_OK;
}

int mi355_bam_fwd(const float* x, const float* const* p, float* y, int B, int C, int Cr, int H, int W, int dilation, void* workspace,
                  size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(y != nullptr);
    if (int rc = bam_check(x, p, B, C, Cr, H, W, dilation, workspace, workspace_bytes)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* mean = ws;                                    // (B,C)      avgpool: bam.py:30
    float* cg = mean + fl((size_t)B * C);                // (B,C)      bn(mlp(.)): :31-32
    float* t0 = cg + fl((size_t)B * C);                  // (B,Cr,HW)  conv1: :55
    float* t1 = t0 + fl((size_t)B * Cr * HW);            // (B,Cr,HW)  conv2 stages: :56
    float* sg = t1 + fl((size_t)B * Cr * HW);            // (B,HW)     bn(conv3(.)): :57-58
    if (int rc = bam_gates(x, p, mean, cg, t0, t1, sg, B, C, Cr, H, W, dilation, st)) return rc;
    ApplyArgs g{};
    g.x = x; g.y = y; g.a = cg; g.b = sg; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_BAM>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// Stand-alone gates of BAM: cg (B,C) = ChannelGate.forward before its expand_as (bam.py:28-33), sg (B,HW) = SpatialGate.forward before
// its expand_as (:53-59).  Either output may be null.  Workspace as for mi355_bam_fwd.
int mi355_bam_gates_fwd(const float* x, const float* const* p, float* cg, float* sg, int B, int C, int Cr, int H, int W, int dilation,
                        void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(cg || sg);
    if (int rc = bam_check(x, p, B, C, Cr, H, W, dilation, workspace, workspace_bytes)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* ws = static_cast<float*>(workspace);
    float* mean = ws;
    float* t0 = mean + 2 * fl((size_t)B * C);
    float* t1 = t0 + fl((size_t)B * Cr * HW);
    if (int rc = bam_gates(x, p, mean, cg, t0, t1, sg, B, C, Cr, H, W, dilation, st)) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// ZPool.forward (triplet_attention.py:31-36): y (B,2,H,W) = [mean over channels, max over channels] of x (B,C,H,W).
int mi355_zpool_fwd(const float* x, float* y, int B, int C, int H, int W, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0);
    launch_chan_reduce<1, 1>(x, nullptr, nullptr, y, B, C, (long)H * W, 2, static_cast<hipStream_t>(stream));
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

// AttentionGate.forward (triplet_attention.py:38-49): y = x * sigmoid(relu(bn(conv_kxk(ZPool(x))))), x (B,C,H,W); w (2,k,k); affine[0] /
// [1] = BatchNorm (eval) scale / shift with the conv bias folded in.  Workspace: 3 * B * H * W floats (+ alignment).
size_t mi355_attention_gate_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return 4 * 3 * fl((size_t)B * H * W) + 256;
}
int mi355_attention_gate_fwd(const float* x, const float* w, const float* affine, float* y, int B, int C, int H, int W, int ksize,
                             void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w && affine && y && workspace && B > 0 && C > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(ksize >= 1 && ksize <= 15 && (ksize & 1));
    MI355_CHECK_ARG(workspace_bytes >= mi355_attention_gate_workspace_bytes(B, H, W) && aligned16(workspace));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long HW = (long)H * W;
    float* zp = static_cast<float*>(workspace);            // (B,2,HW)
    float* gate = zp + 2 * fl((size_t)B * HW);             // (B,HW)
    launch_chan_reduce<1, 1>(x, nullptr, nullptr, zp, B, C, HW, 2, st);
    launch_gate_conv(zp, zp + HW, 2 * HW, w, affine, gate, B, H, W, ksize, st);
    ApplyArgs g{};
    g.x = x; g.y = y; g.c = gate; g.C = C; g.H = H; g.W = W;
    launch_apply<AP_SPATIAL>(g, B, st);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
