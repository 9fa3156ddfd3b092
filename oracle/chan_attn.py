"""Oracle (test infrastructure): channel / spatial attention family -- SE, ECA, CBAM, DoubleAttention.

All functions take NCHW tensors and raw weight tensors (state_dict values) and return the block output.
``dtype`` selects the arithmetic type (torch.float32 to mimic the reference, torch.float64 for a tight
reference when judging which of two fp32 answers is closer to the truth).
"""
import math
import torch
import torch.nn.functional as TF


def _prep(t, dtype):
    return t.detach().to("cpu", dtype)


def se_forward(x, w1, w2, dtype=torch.float32):
    """SELayer.forward -- attention_mechanisms/se_module.py:29-33 (ctor :19-27).

    p[b,c] = mean_hw x ; g = sigmoid(W2 relu(W1 p)) ; y = x * g.   w1:(C/r,C)  w2:(C,C/r), no biases.
    """
    x, w1, w2 = _prep(x, dtype), _prep(w1, dtype), _prep(w2, dtype)
    b, c, h, w = x.shape
    pooled = x.reshape(b, c, h * w).mean(dim=2)                      # (B,C)
    hidden = torch.clamp_min(pooled @ w1.t(), 0)                     # (B,C/r)
    gate = torch.sigmoid(hidden @ w2.t())                            # (B,C)
    return x * gate[:, :, None, None]


def eca_kernel_size(channels, gamma=2, b=1):
    """Kernel-size rule of ECALayer.__init__ -- attention_mechanisms/eca.py:21-22."""
    t = int(abs((math.log(channels, 2) + b) / gamma))
    return t if t % 2 else t + 1


def eca_forward(x, wconv, dtype=torch.float32):
    """ECALayer.forward -- attention_mechanisms/eca.py:26-30.

    g[b,c] = sigmoid(sum_j w[j] * p[b, c + j - (k-1)/2]) with zero padding; y = x * g.  wconv:(1,1,k).
    """
    x, wk = _prep(x, dtype), _prep(wconv, dtype).reshape(-1)
    b, c, h, w = x.shape
    k = wk.numel()
    pad = (k - 1) // 2
    pooled = x.reshape(b, c, h * w).mean(dim=2)
    z = TF.conv1d(pooled[:, None, :], wk.reshape(1, 1, k), padding=pad)[:, 0]     # the ATen op the reference's Conv1d runs
    return x * torch.sigmoid(z)[:, :, None, None]


def eca_gate_explicit(pooled, wk):
    """Tap-by-tap restatement of the k-tap channel conv (cross-correlation, zero pad); cross-checks conv1d in the tests."""
    b, c = pooled.shape
    k = wk.numel()
    pad = (k - 1) // 2
    padded = torch.zeros(b, c + 2 * pad, dtype=pooled.dtype)
    padded[:, pad:pad + c] = pooled
    z = torch.zeros(b, c, dtype=pooled.dtype)
    for j in range(k):
        z = z + wk[j] * padded[:, j:j + c]
    return z


def cbam_channel_forward(x, w1, w2, dtype=torch.float32):
    """ChannelAttention.forward -- attention_mechanisms/cbam.py:31-35.

    fc = 1x1conv(C->C/r) -> ReLU -> 1x1conv(C/r->C), no bias, applied to the avg- and the max-pooled
    vector, summed *after* the second conv (kept as two W2 products here to follow the reference's
    rounding order), then sigmoid and broadcast multiply.
    """
    x = _prep(x, dtype)
    w1 = _prep(w1, dtype).reshape(w1.shape[0], w1.shape[1])
    w2 = _prep(w2, dtype).reshape(w2.shape[0], w2.shape[1])
    b, c, h, w = x.shape
    flat = x.reshape(b, c, h * w)
    avg = flat.mean(dim=2)
    mx = flat.amax(dim=2)
    fa = torch.clamp_min(avg @ w1.t(), 0) @ w2.t()
    fm = torch.clamp_min(mx @ w1.t(), 0) @ w2.t()
    gate = torch.sigmoid(fa + fm)
    return x * gate[:, :, None, None]


def cbam_spatial_forward(x, wconv, dtype=torch.float32):
    """SpatialAttention.forward -- attention_mechanisms/cbam.py:43-48.

    s = [mean_c x, max_c x] (that order, :46) -> KxK cross-correlation 2->1, zero pad K//2, no bias ->
    sigmoid -> multiply.  wconv:(1,2,K,K).
    """
    x, wk = _prep(x, dtype), _prep(wconv, dtype)
    b, c, h, w = x.shape
    ks = wk.shape[-1]
    pad = ks // 2
    smap = torch.stack([x.mean(dim=1), x.amax(dim=1)], dim=1)        # (B,2,H,W)
    acc = TF.conv2d(smap, wk, padding=pad)[:, 0]                      # the ATen op the reference's Conv2d runs
    return x * torch.sigmoid(acc)[:, None, :, :]


def spatial_conv_explicit(smap, wk):
    """Tap-by-tap restatement of the KxK 2->1 conv (cross-correlation, zero pad K//2); cross-checks conv2d in the tests."""
    b, _, h, w = smap.shape
    ks = wk.shape[-1]
    pad = ks // 2
    padded = torch.zeros(b, 2, h + 2 * pad, w + 2 * pad, dtype=smap.dtype)
    padded[:, :, pad:pad + h, pad:pad + w] = smap
    acc = torch.zeros(b, h, w, dtype=smap.dtype)
    for ch in range(2):
        for dy in range(ks):
            for dx in range(ks):
                acc = acc + wk[0, ch, dy, dx] * padded[:, ch, dy:dy + h, dx:dx + w]
    return acc


def cbam_forward(x, w1, w2, wconv, dtype=torch.float32):
    """CBAM.forward -- attention_mechanisms/cbam.py:56-59: spatial stage consumes the channel stage output."""
    return cbam_spatial_forward(cbam_channel_forward(x, w1, w2, dtype), wconv, dtype)


def double_attention_forward(x, wA, bA, wB, bB, wV, bV, wP, bP, dtype=torch.float32):
    """DoubleAttention.forward -- attention_mechanisms/double_attention.py:32-48.

    A = WA X + bA (c_m x HW); Bm = softmax_HW(WB X + bB) (c_n x HW); V = softmax_{c_n}(WV X + bV);
    G = A Bm^T (c_m x c_n); Z = G V (c_m x HW); y = WP Z + bP (C x HW).
    """
    x = _prep(x, dtype)
    b, c, h, w = x.shape
    X = x.reshape(b, c, h * w)

    def pw(wt, bs):                                                  # 1x1 conv as a channel GEMM
        wt = _prep(wt, dtype).reshape(wt.shape[0], wt.shape[1])
        return torch.einsum("oc,bcn->bon", wt, X) + _prep(bs, dtype)[None, :, None]

    A = pw(wA, bA)
    Bm = torch.softmax(pw(wB, bB), dim=2)
    V = torch.softmax(pw(wV, bV), dim=1)
    G = torch.einsum("bmn,bkn->bmk", A, Bm)                          # (B,c_m,c_n)
    Z = torch.einsum("bmk,bkn->bmn", G, V)                           # (B,c_m,HW)
    wp = _prep(wP, dtype).reshape(wP.shape[0], wP.shape[1])
    out = torch.einsum("om,bmn->bon", wp, Z) + _prep(bP, dtype)[None, :, None]
    return out.reshape(b, wp.shape[0], h, w)


# ---- the rest of the channel-attention zoo -------------------------------------------------------------------
def simam_forward(x, e_lambda=1e-4, dtype=torch.float32):
    """simam_module.forward -- attention_mechanisms/simam.py:32-41: d = (x - mean_hw)^2; y = d / (4 (sum_hw d / (HW-1) + lambda)) + 0.5;
    out = x * sigmoid(y)."""
    x = x.detach().to("cpu", dtype)
    n = x.shape[2] * x.shape[3] - 1
    d = (x - x.mean(dim=(2, 3), keepdim=True)) ** 2
    y = d / (4 * (d.sum(dim=(2, 3), keepdim=True) / n + e_lambda)) + 0.5
    return x * torch.sigmoid(y)


def srm_forward(x, cfc, bn_w, bn_b, bn_mean, bn_var, bn_eps=1e-5, dtype=torch.float32):
    """SRM.forward -- attention_mechanisms/srm.py:23-34: u = [mean_hw, std_hw (unbiased)]; z_c = cfc[c,0,0]*mean + cfc[c,0,1]*std (depth-wise
    Conv1d k=2 == per-channel dot product); BatchNorm1d in eval mode; g = sigmoid; out = x * g."""
    x = x.detach().to("cpu", dtype)
    b, c = x.shape[:2]
    flat = x.reshape(b, c, -1)
    mean = flat.mean(-1)
    std = torch.sqrt(((flat - mean[..., None]) ** 2).sum(-1) / (flat.shape[-1] - 1))
    w = cfc.detach().to("cpu", dtype).reshape(c, 2)
    z = w[:, 0] * mean + w[:, 1] * std
    t = lambda v: v.detach().to("cpu", dtype)
    z = (z - t(bn_mean)) / torch.sqrt(t(bn_var) + bn_eps) * t(bn_w) + t(bn_b)
    return x * torch.sigmoid(z)[:, :, None, None]


def _channel_norm(y, eps):
    """(y - mean) / sqrt(E[y^2] - mean^2 + eps) over the last axis (gct.py:25-28, lct.py:31-34)."""
    mean = y.mean(dim=-1, keepdim=True)
    var = (y ** 2).mean(dim=-1, keepdim=True) - mean ** 2
    return (y - mean) / torch.sqrt(var + eps)


def gct_gauss_forward(x, c=2, eps=1e-5, dtype=torch.float32):
    """GCT.forward -- attention_mechanisms/gct.py:23-30: channel means normalised over the channel axis, gate exp(-(yn^2 / 2 * c))."""
    x = x.detach().to("cpu", dtype)
    yn = _channel_norm(x.mean(dim=(2, 3)), eps)
    return x * torch.exp(-(yn ** 2 / 2 * c))[:, :, None, None]


def lct_forward(x, w, b, groups, eps=1e-5, dtype=torch.float32):
    """LCT.forward -- attention_mechanisms/lct.py:29-39: channel means normalised inside each of `groups` channel groups, affine, sigmoid."""
    x = x.detach().to("cpu", dtype)
    B, C = x.shape[:2]
    yn = _channel_norm(x.mean(dim=(2, 3)).reshape(B, groups, -1), eps).reshape(B, C)
    g = torch.sigmoid(w.detach().to("cpu", dtype) * yn + b.detach().to("cpu", dtype))
    return x * g[:, :, None, None]


def gct_forward(x, alpha, gamma, beta, epsilon=1e-5, mode="l2", after_relu=False, dtype=torch.float32):
    """GCT.forward -- attention_mechanisms/gate_channel_module.py:32-50 (l2: :34-36, l1: :38-44): gate = 1 + tanh(e * norm + beta)."""
    x = x.detach().to("cpu", dtype)
    al, ga, be = (p.detach().to("cpu", dtype).reshape(1, -1) for p in (alpha, gamma, beta))
    if mode == "l2":
        e = torch.sqrt((x ** 2).sum(dim=(2, 3)) + epsilon) * al
        norm = ga / torch.sqrt((e ** 2).mean(dim=1, keepdim=True) + epsilon)
    else:
        e = (x if after_relu else x.abs()).sum(dim=(2, 3)) * al
        norm = ga / (e.abs().mean(dim=1, keepdim=True) + epsilon)
    return x * (1.0 + torch.tanh(e * norm + be))[:, :, None, None]


def se_ex_forward(x, w1, b1, w2, b2, gate="sigmoid", dtype=torch.float32):
    """SE with biases and a selectable gate -- cnns/efficientnet.py:23-28 (Linear + bias, sigmoid), cnns/ghostnet.py:59-65 (1x1 conv + bias,
    hard_sigmoid = relu6(z + 3) / 6, :41-45).  b1 / b2 may be None."""
    x = x.detach().to("cpu", dtype)
    t = lambda v: None if v is None else v.detach().to("cpu", dtype)
    w1, w2 = t(w1).reshape(w1.shape[0], -1), t(w2).reshape(w2.shape[0], -1)
    p = x.mean(dim=(2, 3))
    h = p @ w1.t()
    if b1 is not None:
        h = h + t(b1)
    z = torch.relu(h) @ w2.t()
    if b2 is not None:
        z = z + t(b2)
    g = torch.sigmoid(z) if gate == "sigmoid" else torch.clamp(z + 3.0, 0.0, 6.0) / 6.0
    return x * g[:, :, None, None]
