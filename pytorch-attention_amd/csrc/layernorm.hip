// layernorm.hip -- nn.LayerNorm over the last axis (eps inside the sqrt, biased variance), fp32, one wave per row.
// HBM-bound: a row (<= 4 KiB at C <= 1024) lives in registers between the statistics and the normalise pass, so each
// element is read once and written once.  Mean and variance are two-pass (sum, then sum of squared deviations), the
// same formulation as the reference's nn.LayerNorm (ViT.py:111-114, cswin.py:139,174, xcit.py:271-283, mlp_mixer.py:39-41).
#include "common.h"

namespace {
using v4f = float __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void layernorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, long rows,
                                                           int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int n4 = cols >> 2;
    const float inv = 1.0f / (float)cols;
    for (long row = wave0; row < rows; row += nwaves) {
        const v4f* xr = reinterpret_cast<const v4f*>(x + row * cols);
        v4f v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            v[j] = (i < n4) ? xr[i] : v4f{0.f, 0.f, 0.f, 0.f};
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < n4) {
                const v4f d = v[j] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        v4f* yr = reinterpret_cast<v4f*>(y + row * cols);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < n4) {
                const v4f ww = reinterpret_cast<const v4f*>(w)[i], bb = reinterpret_cast<const v4f*>(b)[i];
                yr[i] = (v[j] - mean) * rstd * ww + bb;
            }
        }
    }
}

// any width / alignment: three sweeps over the (cache-resident) row
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, long rows,
                                                               int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long row = wave0; row < rows; row += nwaves) {
        const float* xr = x + row * cols;
        float s = 0.f;
        for (int i = lane; i < cols; i += 64) s += xr[i];
        const float mean = wave_sum(s) / (float)cols;
        float q = 0.f;
        for (int i = lane; i < cols; i += 64) { const float d = xr[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
        for (int i = lane; i < cols; i += 64) y[row * cols + i] = (xr[i] - mean) * rstd * w[i] + b[i];
    }
}
}  // namespace

extern "C" int mi355_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, int rows, int cols, float eps,
                                   mi355_stream_t stream) {
    MI355_CHECK_ARG(x && weight && bias && y && rows > 0 && cols > 0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192;
    const bool vec = (cols % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(weight) && aligned16(bias);
    if (vec && cols <= 256)       layernorm_reg_kernel<1><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps);
    else if (vec && cols <= 512)  layernorm_reg_kernel<2><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps);
    else if (vec && cols <= 1024) layernorm_reg_kernel<4><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps);
    else if (vec && cols <= 2048) layernorm_reg_kernel<8><<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps);
    else                          layernorm_generic_kernel<<<grid, 256, 0, st>>>(x, weight, bias, y, rows, cols, eps);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}
