"""Oracle (test infrastructure): XCiT cross-covariance attention (XCA), LPI and XCABlock (eval mode)."""
import torch
from .transformer import layernorm, gelu, linear, _t, _sub


def xca_forward(x, p, num_heads, dtype=torch.float32):
    """XCA.forward -- vision_transformers/xcit.py:245-265.

    Per head: q,k,v as (d x N); q,k L2-normalised along N (F.normalize, eps 1e-12 on the norm);
    A = softmax_rows((q k^T) * temperature_h) (d x d); O = A v (d x N); y[b,n,i*d+j] = O_i[j,n]; proj.
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    qkv = linear(x, _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    temp = _t(p["temperature"], dtype).reshape(num_heads)
    out = torch.empty(B, N, C, dtype=dtype)
    for i in range(num_heads):
        q = qkv[:, :, 0 * C + i * d: 0 * C + (i + 1) * d].transpose(1, 2)      # (B,d,N)
        k = qkv[:, :, 1 * C + i * d: 1 * C + (i + 1) * d].transpose(1, 2)
        v = qkv[:, :, 2 * C + i * d: 2 * C + (i + 1) * d].transpose(1, 2)
        qn = q / torch.clamp_min(torch.sqrt((q * q).sum(dim=2, keepdim=True)), 1e-12)
        kn = k / torch.clamp_min(torch.sqrt((k * k).sum(dim=2, keepdim=True)), 1e-12)
        a = (qn @ kn.transpose(1, 2)) * temp[i]                                # (B,d,d)
        a = torch.softmax(a, dim=2)
        out[:, :, i * d:(i + 1) * d] = (a @ v).transpose(1, 2)
    return linear(out, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def _dw3x3(img, w, b):
    """Depth-wise 3x3 cross-correlation, zero pad 1.  img (B,C,H,W), w (C,1,3,3), b (C,)."""
    B, C, H, W = img.shape
    halo = torch.zeros(B, C, H + 2, W + 2, dtype=img.dtype)
    halo[:, :, 1:1 + H, 1:1 + W] = img
    acc = torch.zeros_like(img) + b[None, :, None, None]
    for dy in range(3):
        for dx in range(3):
            acc = acc + halo[:, :, dy:dy + H, dx:dx + W] * w[None, :, 0, dy, dx, None, None]
    return acc


def lpi_forward(x, p, H, W, dtype=torch.float32, bn_eps=1e-5):
    """LPI.forward -- vision_transformers/xcit.py:149-157, BatchNorm2d in eval mode (running stats).

    tokens (B,N,C) -> image (B,C,H,W) -> dw3x3 -> GELU -> BN(eval) -> dw3x3 -> tokens.
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    img = x.transpose(1, 2).reshape(B, C, H, W)
    y = gelu(_dw3x3(img, _t(p["conv1.weight"], dtype), _t(p["conv1.bias"], dtype)))
    mean, var = _t(p["bn.running_mean"], dtype), _t(p["bn.running_var"], dtype)
    g, bt = _t(p["bn.weight"], dtype), _t(p["bn.bias"], dtype)
    y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn_eps) \
        * g[None, :, None, None] + bt[None, :, None, None]
    y = _dw3x3(y, _t(p["conv2.weight"], dtype), _t(p["conv2.bias"], dtype))
    return y.reshape(B, C, N).transpose(1, 2)


def xca_block_forward(x, p, num_heads, H, W, dtype=torch.float32):
    """XCABlock.forward -- vision_transformers/xcit.py:290-294.

    x += gamma1 * XCA(LN1 x); x += gamma3 * LPI(LN3 x); x += gamma2 * Mlp(LN2 x)   (Mlp :32-38, one GELU).
    """
    x = _t(x, dtype)
    u = layernorm(x, _t(p["norm1.weight"], dtype), _t(p["norm1.bias"], dtype))
    x = x + _t(p["gamma1"], dtype) * xca_forward(u, _sub(p, "attn."), num_heads, dtype)
    u = layernorm(x, _t(p["norm3.weight"], dtype), _t(p["norm3.bias"], dtype))
    x = x + _t(p["gamma3"], dtype) * lpi_forward(u, _sub(p, "local_mp."), H, W, dtype)
    u = layernorm(x, _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
    m = _sub(p, "mlp.")
    h = gelu(linear(u, _t(m["fc1.weight"], dtype), _t(m["fc1.bias"], dtype)))
    return x + _t(p["gamma2"], dtype) * linear(h, _t(m["fc2.weight"], dtype), _t(m["fc2.bias"], dtype))
