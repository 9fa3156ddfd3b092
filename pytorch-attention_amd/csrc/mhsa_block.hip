// mhsa_block.hip -- ViT Attention.forward (ViT.py:79-89) as ONE C call: cast -> qkv GEMM (one launch at C = 256 / 384 / 512) -> attention core ->
// proj GEMM (+ residual).
// SURVEY 8(b) lists `mhsa` among the ops the boundary exports; a host that is not Python gets the block without re-implementing
// the dispatch rules of mi355attn/modules/vit.py.  Nothing new runs on the device: the entry composes the library's own entry
// points on the caller's stream, with q / k / v / context in the caller's workspace in the 16-bit operand format.
#include "common.h"

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" {

size_t mi355_mhsa_workspace_bytes(int B, int N, int C, int x_is16) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    const size_t M = (size_t)B * N;
    size_t lin = mi355_linear16_workspace_bytes((int)M, 3 * C, C);
    const size_t lin2 = mi355_linear16_workspace_bytes((int)M, C, C);
    if (lin2 > lin) lin = lin2;
    return (x_is16 ? 0 : up256(M * C * 2)) + up256(M * 3 * C * 2) + up256(M * C * 2) + up256(lin) + 256;
}

int mi355_mhsa_fwd(const void* x, int x_is16, const void* Wqkv16, const float* b_qkv, const void* Wproj16, const float* b_proj,
                   const float* resid, float* y, int B, int N, int C, int heads, float scale, int precision, void* workspace,
                   size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && Wqkv16 && Wproj16 && y && workspace && B > 0 && N > 0 && C > 0 && heads > 0);
    MI355_CHECK_ARG(precision == MI355_PREC_FP16 || precision == MI355_PREC_BF16);
    MI355_CHECK_ARG(x_is16 == 0 || x_is16 == 1);
    MI355_CHECK_ARG(workspace_bytes >= mi355_mhsa_workspace_bytes(B, N, C, x_is16) && aligned16(workspace));
    if ((C % heads) || (C % 64))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_mhsa_fwd: C %% heads == 0 and C %% 64 == 0 (C=%d heads=%d)", C, heads);
    const int d = C / heads;
    if (!(d == 32 || d == 64 || d == 128 || d == 192 || d == 256))
        return mi355::fail(MI355_EUNSUPPORTED, "mi355_mhsa_fwd: head_dim %d (built: 32, 64, 128, 192, 256)", d);
    const size_t M = (size_t)B * N;
    if (M > (size_t)0x7fffffff) return mi355::fail(MI355_EUNSUPPORTED, "mi355_mhsa_fwd: B * N too large");
    char* w = static_cast<char*>(workspace);
    void* xb = nullptr;
    if (!x_is16) {
        xb = w;
        w += up256(M * C * 2);
    }
    void* qkv16 = w;
    w += up256(M * 3 * C * 2);
    void* ctx16 = w;
    w += up256(M * C * 2);
    void* lws = w;
    size_t lbytes = mi355_linear16_workspace_bytes((int)M, 3 * C, C);
    // fp32 x at C = 256 / 384 / 512: the cast rides in the qkv GEMM's staging (round 6, mi355_linear16_x32_fwd: one launch, no 16-bit copy of x);
    // MI355_EUNSUPPORTED there = nothing launched: cast, then the 16-bit entry
    int rc_qkv = MI355_EUNSUPPORTED;
    if (!x_is16)
        rc_qkv = mi355_linear16_x32_fwd(static_cast<const float*>(x), Wqkv16, b_qkv, qkv16, (int)M, 3 * C, C, C, 3 * C, MI355_ACT_NONE, precision, stream);
    if (rc_qkv == MI355_EUNSUPPORTED) {
        const void* x16 = x;
        if (!x_is16) {
            if (int rc = mi355_cast16_fwd(static_cast<const float*>(x), xb, M * C, precision, stream)) return rc;
            x16 = xb;
        }
        rc_qkv = mi355_linear16_ws_fwd(x16, Wqkv16, b_qkv, nullptr, nullptr, qkv16, (int)M, 3 * C, C, C, 3 * C, MI355_ACT_NONE, 1, precision,
                                       lbytes ? lws : nullptr, lbytes, stream);
    }
    if (rc_qkv) return rc_qkv;
    if ((d == 32 || d == 64) && N <= 224) {
        if (int rc = mi355_sdpa16_fwd(qkv16, ctx16, B, N, heads, d, scale, precision, stream)) return rc;
    } else {
        const char* q = static_cast<const char*>(qkv16);
        if (int rc = mi355_sdpa_general_fwd(q, q + (size_t)C * 2, q + (size_t)2 * C * 2, nullptr, ctx16, B, heads, N, N, d, 3L * C, 3L * C, 3L * C,
                                            (long)C, 0, scale, 1, precision, stream))
            return rc;
    }
    lbytes = mi355_linear16_workspace_bytes((int)M, C, C);
    return mi355_linear16_ws_fwd(ctx16, Wproj16, b_proj, nullptr, resid, y, (int)M, C, C, C, C, MI355_ACT_NONE, 0, precision,
                                 lbytes ? lws : nullptr, lbytes, stream);
}

}  // extern "C"
