// stem_conv.hip -- direct fp32 convolution for the FIRST layer of a conv stem on the NCHW image (XCiT ConvPatchEmbed xcit.py:88-126:
// 3 -> 16 channels, 3x3 stride 2, 224^2 -> 112^2), written token-major with bias (+ position rows) and GELU fused.
//
// That layer is HBM-shaped, not GEMM-shaped: K = Cin*kh*kw = 27 and N = Cout = 16, so the implicit-GEMM path of gemm.hip fills a
// 128-wide MFMA tile with 16 useful columns and spends its time gathering (707 us per XCiT-nano forward at B = 256 against ~65 us
// of traffic).  Here a thread owns two horizontally adjacent output pixels and all COUT output channels in registers; the weights
// sit in LDS transposed to [k][COUT] and come in as broadcast 16-byte reads (one read feeds eight FMAs); the image is read in
// place, neighbours sharing it through L1/L2; a thread writes COUT contiguous floats per pixel: 200 us.  Exact fp32 arithmetic in
// every precision mode.  The same kernel on the token-major second layer (16 -> 32, K = 144) measured 621 us against 231 us for the
// implicit GEMM -- wider layers stay on the MFMA path.
#include "common.h"
#include "mma.h"

namespace mi355 {

namespace {

struct StemArgs {
    const float* x; const float* w; const float* bias; const float* pos; float* y;
    int Cin, H, W, Cout, KH, KW, stride, pad, ldw, OH, OW, act;
    long npair;          // B * OH * ceil(OW / 2)
};

template <int COUT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [K][COUT], zero past Cout
    const int K = a.Cin * a.KH * a.KW;
    for (int q = threadIdx.x; q < K * COUT; q += 256) {
        const int k = q / COUT, co = q - k * COUT;
        wl[q] = co < a.Cout ? a.w[(long)co * a.ldw + k] : 0.f;
    }
    __syncthreads();
    const long pid = (long)blockIdx.x * 256 + threadIdx.x;
    if (pid >= a.npair) return;
    const int ow2 = (a.OW + 1) >> 1;
    const int oxp = (int)(pid % ow2), oy = (int)((pid / ow2) % a.OH);
    const long b = pid / ((long)ow2 * a.OH);
    const int ox0 = oxp * 2;
    const bool has1 = ox0 + 1 < a.OW;
    f4 acc0[COUT / 4], acc1[COUT / 4];
#pragma unroll
    for (int c = 0; c < COUT / 4; ++c) { acc0[c] = f4{0.f, 0.f, 0.f, 0.f}; acc1[c] = f4{0.f, 0.f, 0.f, 0.f}; }
    const long HW = (long)a.H * a.W;
    const float* xb = a.x + b * a.Cin * HW;
    auto fma_k = [&](int k, float v0, float v1) {
        const f4* wr = reinterpret_cast<const f4*>(wl + k * COUT);
#pragma unroll
        for (int c = 0; c < COUT / 4; ++c) {
            const f4 wv = wr[c];
            acc0[c] += wv * v0;
            acc1[c] += wv * v1;
        }
    };
    {                                                            // K ordered (c, ky, kx)
        int k = 0;
        for (int c = 0; c < a.Cin; ++c)
            for (int ky = 0; ky < a.KH; ++ky) {
                const int iy = oy * a.stride - a.pad + ky;
                const bool rowin = iy >= 0 && iy < a.H;
                const float* row = xb + (long)c * HW + (long)(rowin ? iy : 0) * a.W;
                for (int kx = 0; kx < a.KW; ++kx, ++k) {
                    const int ix0 = ox0 * a.stride - a.pad + kx, ix1 = ix0 + a.stride;
                    const float v0 = (rowin && ix0 >= 0 && ix0 < a.W) ? row[ix0] : 0.f;
                    const float v1 = (rowin && has1 && ix1 >= 0 && ix1 < a.W) ? row[ix1] : 0.f;
                    fma_k(k, v0, v1);
                }
            }
    }
    const long P = (long)a.OH * a.OW;
    const long p0 = (long)oy * a.OW + ox0;
    float* y0 = a.y + (b * P + p0) * a.Cout;
#pragma unroll
    for (int c = 0; c < COUT / 4; ++c) {
        if (c * 4 >= a.Cout) continue;
        f4 bv = a.bias ? *reinterpret_cast<const f4*>(a.bias + c * 4) : f4{0.f, 0.f, 0.f, 0.f};
        f4 r0 = acc0[c] + bv, r1 = acc1[c] + bv;
        if (a.pos) {
            r0 += *reinterpret_cast<const f4*>(a.pos + p0 * a.Cout + c * 4);
            if (has1) r1 += *reinterpret_cast<const f4*>(a.pos + (p0 + 1) * a.Cout + c * 4);
        }
        if (a.act == MI355_ACT_GELU) {
            r0 = gelu_fast4(r0);
            r1 = gelu_fast4(r1);
        }
        *reinterpret_cast<f4*>(y0 + c * 4) = r0;
        if (has1) *reinterpret_cast<f4*>(y0 + a.Cout + c * 4) = r1;
    }
}

}  // namespace

// Envelope: the image layer of a stem -- NCHW input with <= 4 channels, narrow output (see the header).
bool stem_conv_applicable(int Cin, int Cout, int KH, int KW, int in_layout, const float* bias, const float* pos, const float* y) {
    const long work = (long)Cin * KH * KW * Cout;                // FMAs per output pixel
    if (in_layout != 0 || Cin > 4 || Cout > 64 || (Cout & 3) || work > 2048) return false;
    return aligned16(y) && (!bias || aligned16(bias)) && (!pos || aligned16(pos));
}

int stem_conv(const float* x, const float* w, const float* bias, const float* pos, float* y, int B, int Cin, int H, int W, int Cout, int KH,
              int KW, int stride, int pad, int ldw, int in_layout, int act, hipStream_t st) {
    StemArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.pos = pos; a.y = y;
    a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.ldw = ldw; a.act = act;
    a.OH = (H + 2 * pad - KH) / stride + 1; a.OW = (W + 2 * pad - KW) / stride + 1;
    a.npair = (long)B * a.OH * ((a.OW + 1) / 2);
    const int grid = cdiv(a.npair, 256);
    const int K = Cin * KH * KW;
#define STEM(COUT_)                                                                                                   \
    do {                                                                                                              \
        const size_t lds = (size_t)K * COUT_ * sizeof(float);                                                         \
        stem_conv_kernel<COUT_><<<grid, 256, lds, st>>>(a);                                                           \
    } while (0)
    if (Cout <= 16) STEM(16); else if (Cout <= 32) STEM(32); else STEM(64);
#undef STEM
    return MI355_OK;
}

}  // namespace mi355
