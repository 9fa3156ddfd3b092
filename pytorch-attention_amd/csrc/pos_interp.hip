// pos_interp.hip -- bicubic resize of a ViT position-embedding table (ViT.py:160-178 interpolate_pos_encoding).
//
// The reference reshapes the (n0*n0, dim) patch rows of the table to (1, dim, n0, n0) and calls
// F.interpolate(scale_factor=(sf_h, sf_w), mode='bicubic') (align_corners = False): output pixel (oy, ox) samples the source at
// (oy + 0.5) / sf_h - 0.5 (no clamp of the coordinate for the cubic filter), four taps per axis with the cubic-convolution kernel
// A = -0.75, tap indices clamped to the table.  Run once per (resolution, parameter version) by the host mirror and cached; rows stay
// token-major, so no permute is needed on either side.
#include "common.h"

namespace {

__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x1 = t, x2 = 1.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void bicubic_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0h, int n0w,
                                                           int oh, int ow, int dim, float rh, float rw) {
    const int token = blockIdx.x;
    const int oy = token / ow, ox = token - oy * ow;
    const float ry = rh * ((float)oy + 0.5f) - 0.5f, rx = rw * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coeffs(ry - fy, cy);
    cubic_coeffs(rx - fx, cx);
    for (int c = threadIdx.x; c < dim; c += 256) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), n0h - 1);
            float row = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), n0w - 1);
                row += cx[j] * src[((long)yy * n0w + xx) * dim + c];
            }
            acc += cy[i] * row;
        }
        dst[(long)token * dim + c] = acc;
    }
}

}  // namespace

extern "C" int mi355_bicubic_rows_fwd(const float* table, float* out, int n0h, int n0w, int oh, int ow, int dim, float scale_h,
                                      float scale_w, mi355_stream_t stream) {
    MI355_CHECK_ARG(table && out && n0h > 0 && n0w > 0 && oh > 0 && ow > 0 && dim > 0 && scale_h > 0.f && scale_w > 0.f);
    bicubic_rows_kernel<<<oh * ow, 256, 0, static_cast<hipStream_t>(stream)>>>(table, out, n0h, n0w, oh, ow, dim, 1.0f / scale_h,
                                                                              1.0f / scale_w);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}
