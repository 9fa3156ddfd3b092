// double_attn.hip -- A2-Net DoubleAttention forward (double_attention.py:32-48) on the MFMA GEMM engine.
//
//   [A; Bm; V] = [WA; WB; WV] X_b + bias          one batched K-major GEMM (the three 1x1 convs share the read of X)
//   Bm = softmax over HW (rows),  V = softmax over c_n (columns)      two small fp32 kernels, in place
//   G = A Bm^T (c_m x c_n, contraction over HW)    batched NT GEMM
//   M' = WP G  (C x c_n, tiny)                     batched K-major GEMM;  y = WP (G V) + bP is evaluated as (WP G) V + bP
//   y = M' V + bP                                   batched K-major GEMM with row bias
// x, y are NCHW, i.e. each image is a (C x HW) matrix with HW contiguous: the activation is always the K-major operand
// of the engine, never transposed in memory.
#include "common.h"

namespace {

// in-place softmax over the last axis of `rows` rows; row r of image b starts at base + b*img_stride + r*cols
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ p, int rows_per_img, int cols, long img_stride,
                                                          long total_rows) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (long r = wave0; r < total_rows; r += nw) {
        float* row = p + (r / rows_per_img) * img_stride + (r % rows_per_img) * (long)cols;
        float m = -INFINITY;
        for (int i = lane; i < cols; i += 64) m = fmaxf(m, row[i]);
        m = wave_max(m);
        float s = 0.f;
        for (int i = lane; i < cols; i += 64) s += expf(row[i] - m);
        s = wave_sum(s);
        for (int i = lane; i < cols; i += 64) row[i] = expf(row[i] - m) / s;
    }
}

// The same with the row held in registers (one read + one write of the tensor instead of three reads + one write): cols % 4 == 0,
// cols <= 256 * NV4, one wave per row, NV4 16-byte groups per lane.
template <int NV4>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* __restrict__ p, int rows_per_img, int cols, long img_stride, long total_rows) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total_rows) return;
    float* row = p + (r / rows_per_img) * img_stride + (r % rows_per_img) * (long)cols;
    const int n4 = cols >> 2;
    f4_t v[NV4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < n4 ? *reinterpret_cast<const f4_t*>(row + i * 4) : f4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        v[k] = f4_t{expf(v[k].x - m), expf(v[k].y - m), expf(v[k].z - m), expf(v[k].w - m)};
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    s = wave_sum(s);
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int i = lane + 64 * k;
        if (i < n4) *reinterpret_cast<f4_t*>(row + i * 4) = v[k] / s;
    }
}

// in-place softmax over the channel axis: element (b, j, p) at base + b*img_stride + j*HW + p, j < cn
__global__ __launch_bounds__(256) void softmax_cols_kernel(float* __restrict__ p, int cn, int HW, long img_stride, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float* col = p + (i / HW) * img_stride + (i % HW);
    float m = -INFINITY;
    for (int j = 0; j < cn; ++j) m = fmaxf(m, col[(long)j * HW]);
    float s = 0.f;
    for (int j = 0; j < cn; ++j) s += expf(col[(long)j * HW] - m);
    for (int j = 0; j < cn; ++j) col[(long)j * HW] = expf(col[(long)j * HW] - m) / s;
}

// The same softmax with the column held in registers: 256 threads = 64 pixels x 4 channel quarters, CPT channels per thread
// (one read + one write of the tensor instead of three reads + one write).
template <int CPT>
__global__ __launch_bounds__(256) void softmax_cols_reg_kernel(float* __restrict__ p, int cn, int HW, long img_stride, long total) {
    __shared__ float s_m[4][64], s_s[4][64];
    const int px = threadIdx.x & 63, part = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + px;
    const bool live = i < total;
    float* col = p + (live ? (i / HW) * img_stride + (i % HW) : 0) + (long)part * CPT * HW;
    float v[CPT];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        v[j] = (live && part * CPT + j < cn) ? col[(long)j * HW] : -INFINITY;
        m = fmaxf(m, v[j]);
    }
    s_m[part][px] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0][px], s_m[1][px]), fmaxf(s_m[2][px], s_m[3][px]));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        v[j] = expf(v[j] - m);                                   // exp(-inf) = 0 for the padding channels
        s += v[j];
    }
    s_s[part][px] = s;
    __syncthreads();
    s = (s_s[0][px] + s_s[1][px]) + (s_s[2][px] + s_s[3][px]);
    if (live) {
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (part * CPT + j < cn) col[(long)j * HW] = v[j] / s;
    }
}

inline size_t r16(size_t n) { return (n + 15) & ~(size_t)15; }

// x (B, C, HW) fp32 -> xt (B, HW, C) in the 16-bit operand type: the token-major activation the fast GEMM engine wants as its row operand
// (64 x 64 tiles through LDS: 256-byte runs in, 128-byte runs out).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_tokens16_kernel(const float* __restrict__ x, T* __restrict__ xt, int C, int HW) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, t = threadIdx.x;
    {
        const int tx = t & 63, ty = t >> 6;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = ty * 16 + r, c = c0 + cl, p = p0 + tx;
            tile[cl][tx] = (c < C && p < HW) ? x[((long)b * C + c) * HW + p] : 0.f;
        }
    }
    __syncthreads();
    const int pl = t >> 2, cq = t & 3, p = p0 + pl;
    if (p >= HW) return;
    T* dst = xt + ((long)b * HW + p) * C + c0 + cq * 16;
    if (c0 + cq * 16 + 16 <= C && (C & 7) == 0) {                 // two 16-byte stores of eight elements
        typedef T v8t __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v8t o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (T)tile[cq * 16 + h * 8 + k][pl];
            *reinterpret_cast<v8t*>(dst + h * 8) = o;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (c0 + cq * 16 + k < C) dst[k] = (T)tile[cq * 16 + k][pl];
    }
}

}  // namespace

extern "C" {

// workspace: Wcat (M3 x C) | bcat (M3) | ABV (B, M3, HW) | G (B, cm, cn) | M' (B, C, cn)
size_t mi355_double_attn_workspace_bytes(int B, int C, int cm, int cn, int H, int W) {
    const size_t M3 = (size_t)cm + 2 * (size_t)cn, HW = (size_t)H * W;
    return r16(M3 * C * 4) + r16(M3 * 4) + r16((size_t)B * M3 * HW * 4) + r16((size_t)B * cm * cn * 4) + r16((size_t)B * C * cn * 4) +
           r16((size_t)B * HW * C * 2) + r16(M3 * C * 2);      // token-major 16-bit copy of x and the 16-bit weights of the fast first product
}

// The same for one precision mode: the two-pass path of the 16-bit modes (double_attn_fused.hip) needs the 16-bit V tensor and the
// per-range partial results only -- 0.2 GB instead of 1.7 GB at (256, 256, 56, 56).
size_t mi355_double_attn_ws_bytes(int B, int C, int cm, int cn, int H, int W, int precision) {
    if (B > 0 && C > 0 && H > 0 && W > 0 && mi355::opt_da_fused() != 0 && mi355::double_attn_fused_ok(B, C, cm, cn, H * W, precision))
        return mi355::double_attn_fused_workspace(B, C, H * W);
    return mi355_double_attn_workspace_bytes(B, C, cm, cn, H, W);
}

int mi355_double_attn_fwd(const float* x, const float* wA, const float* bA, const float* wB, const float* bB, const float* wV,
                          const float* bV, const float* wP, const float* bP, float* y, int B, int C, int cm, int cn, int H, int W,
                          int precision, void* ws, size_t ws_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && wA && bA && wB && bB && wV && bV && wP && bP && y && ws);
    MI355_CHECK_ARG(B > 0 && C > 0 && cm > 0 && cn > 0 && H > 0 && W > 0);
    MI355_CHECK_ARG(ws_bytes >= mi355_double_attn_ws_bytes(B, C, cm, cn, H, W, precision));
    const int HW = H * W, M3 = cm + 2 * cn;
    if (mi355::opt_da_fused() != 0 && mi355::double_attn_small_ok(B, C, cm, cn, HW, precision) && aligned16(x) && aligned16(y) && aligned16(wA) &&
        aligned16(wB) && aligned16(wV) && aligned16(bV))
        return mi355::double_attn_small(x, wA, bA, wB, bB, wV, bV, wP, bP, y, B, C, HW, precision, static_cast<hipStream_t>(stream));
    if (mi355::double_attn_fused_ok(B, C, cm, cn, HW, precision) && aligned16(x) && aligned16(y) && aligned16(ws) && mi355::opt_da_fused() != 0)
        return mi355::double_attn_fused(x, wA, bA, wB, bB, wV, bV, wP, bP, y, B, C, HW, precision, ws, static_cast<hipStream_t>(stream));
    if ((HW & 3) || (C & 3) || (cm & 3) || (cn & 3) || !aligned16(x) || !aligned16(y) || !aligned16(ws))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_double_attn_fwd: H*W, C, c_m, c_n must be multiples of 4 (HW=%d C=%d cm=%d cn=%d)",
                           HW, C, cm, cn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* p = static_cast<char*>(ws);
    float* Wcat = reinterpret_cast<float*>(p); p += r16((size_t)M3 * C * 4);
    float* bcat = reinterpret_cast<float*>(p); p += r16((size_t)M3 * 4);
    float* ABV = reinterpret_cast<float*>(p);  p += r16((size_t)B * M3 * HW * 4);
    float* G = reinterpret_cast<float*>(p);    p += r16((size_t)B * cm * cn * 4);
    float* Z = reinterpret_cast<float*>(p);
    MI355_HIP(hipMemcpyAsync(Wcat, wA, (size_t)cm * C * 4, hipMemcpyDeviceToDevice, st));
    MI355_HIP(hipMemcpyAsync(Wcat + (size_t)cm * C, wB, (size_t)cn * C * 4, hipMemcpyDeviceToDevice, st));
    MI355_HIP(hipMemcpyAsync(Wcat + (size_t)(cm + cn) * C, wV, (size_t)cn * C * 4, hipMemcpyDeviceToDevice, st));
    MI355_HIP(hipMemcpyAsync(bcat, bA, (size_t)cm * 4, hipMemcpyDeviceToDevice, st));
    MI355_HIP(hipMemcpyAsync(bcat + cm, bB, (size_t)cn * 4, hipMemcpyDeviceToDevice, st));
    MI355_HIP(hipMemcpyAsync(bcat + cm + cn, bV, (size_t)cn * 4, hipMemcpyDeviceToDevice, st));
    const long sABV = (long)M3 * HW;
    int rc;
    if (precision != MI355_PREC_STRICT && (C & 63) == 0) {
        // The three 1x1 convs on the 16-bit GEMM engine: x is copied once to token-major 16-bit (the engine's row operand), and the
        // product is written back CHANNEL-major by the transposing epilogue of mi355_linear16_tr_fwd (rows per image = HW) -- the
        // same (B, M3, HW) ABV tensor as the fp32-in path below, at a third of its time.
        char* q = reinterpret_cast<char*>(Z) + r16((size_t)B * C * cn * 4);
        void* xt16 = q; q += r16((size_t)B * HW * C * 2);
        void* w16 = q;
        const dim3 tgrid(cdiv(HW, 64), cdiv(C, 64), B);
        if (precision == MI355_PREC_FP16) nchw_to_tokens16_kernel<_Float16><<<tgrid, 256, 0, st>>>(x, static_cast<_Float16*>(xt16), C, HW);
        else                              nchw_to_tokens16_kernel<__bf16><<<tgrid, 256, 0, st>>>(x, static_cast<__bf16*>(xt16), C, HW);
        rc = mi355_cast16_fwd(Wcat, w16, (size_t)M3 * C, precision, stream);
        if (rc) return rc;
        rc = mi355_linear16_tr_fwd(xt16, w16, bcat, nullptr, ABV, B * HW, M3, C, C, HW, precision, stream);
        if (rc) return rc;
    } else {
        rc = mi355::gemm_kn_batched(Wcat, x, bcat, nullptr, ABV, B, M3, HW, C, C, HW, HW, 0, (long)C * HW, sABV, MI355_ACT_NONE, precision, st);
        if (rc) return rc;
    }
    {
        const long rows = (long)B * cn;
        const int grid = cdiv(rows, 4) < 8192 ? cdiv(rows, 4) : 8192;
        float* Bp = ABV + (long)cm * HW;
        if (HW <= 1024)      softmax_rows_reg_kernel<4><<<cdiv(rows, 4), 256, 0, st>>>(Bp, cn, HW, sABV, rows);      // HW % 4 == 0 checked above
        else if (HW <= 2048) softmax_rows_reg_kernel<8><<<cdiv(rows, 4), 256, 0, st>>>(Bp, cn, HW, sABV, rows);
        else if (HW <= 4096) softmax_rows_reg_kernel<16><<<cdiv(rows, 4), 256, 0, st>>>(Bp, cn, HW, sABV, rows);
        else                 softmax_rows_kernel<<<grid, 256, 0, st>>>(Bp, cn, HW, sABV, rows);
        const long total = (long)B * HW;
        float* Vp = ABV + (long)(cm + cn) * HW;
        if (cn <= 64)       softmax_cols_reg_kernel<16><<<cdiv(total, 64), 256, 0, st>>>(Vp, cn, HW, sABV, total);
        else if (cn <= 128) softmax_cols_reg_kernel<32><<<cdiv(total, 64), 256, 0, st>>>(Vp, cn, HW, sABV, total);
        else if (cn <= 256) softmax_cols_reg_kernel<64><<<cdiv(total, 64), 256, 0, st>>>(Vp, cn, HW, sABV, total);
        else                softmax_cols_kernel<<<cdiv(total, 256), 256, 0, st>>>(Vp, cn, HW, sABV, total);
    }
    rc = mi355::gemm_nt_batched(ABV, ABV + (long)cm * HW, G, B, cm, cn, HW, HW, HW, cn, sABV, sABV, (long)cm * cn, precision, st);
    if (rc) return rc;
    // y = WP (G V) + bP = (WP G) V + bP: the (C x c_n) product per image is tiny, and the (c_m x HW) intermediate Z of the reference's
    // order (a write + a read of B*c_m*HW floats and one more pass of the engine) disappears.
    float* Mp = Z;                                              // (B, C, cn) in the old Z region
    rc = mi355::gemm_kn_batched(wP, G, nullptr, nullptr, Mp, B, C, cn, cm, cm, cn, cn, 0, (long)cm * cn, (long)C * cn, MI355_ACT_NONE,
                                precision, st);
    if (rc) return rc;
    rc = mi355::gemm_kn_batched(Mp, ABV + (long)(cm + cn) * HW, bP, nullptr, y, B, C, HW, cn, cn, HW, HW, (long)C * cn, sABV,
                                (long)C * HW, MI355_ACT_NONE, precision, st);
    if (rc) return rc;
    MI355_LAUNCH_CHECK();
    return MI355_OK;
}

}  // extern "C"
