"""Oracle (test infrastructure): CSWin LePEAttention and CSWinBlock, with explicit window index math."""
import torch
import torch.nn.functional as TF
from .transformer import layernorm, gelu, linear, sdpa_core, _t, _sub


def window_token_index(reso, H_sp, W_sp):
    """Token index table idx[w, t] = l for the stripe partition used by img2windows/windows2img.

    Follows vision_transformers/cswin.py:199-216: token l sits at (y, x) = (l // W, l % W); its window is
    w = (y // H_sp) * (W // W_sp) + x // W_sp and its slot inside the window is
    t = (y % H_sp) * W_sp + x % W_sp.  Returned as an int64 tensor (nWin, H_sp*W_sp).
    """
    H = W = reso
    nW = W // W_sp
    nwin = (H // H_sp) * nW
    idx = torch.empty(nwin, H_sp * W_sp, dtype=torch.int64)
    for l in range(H * W):
        y, x = divmod(l, W)
        idx[(y // H_sp) * nW + x // W_sp, (y % H_sp) * W_sp + x % W_sp] = l
    return idx


def _stripe_shape(reso, idx, split_size):
    """(H_sp, W_sp) rule of LePEAttention.__init__ -- cswin.py:62-67."""
    if idx == -1:
        return reso, reso
    if idx == 0:
        return reso, split_size
    if idx == 1:
        return split_size, reso
    raise ValueError(f"bad LePEAttention idx {idx}")


def lepe_attention_forward(qkv, get_v_w, get_v_b, reso, idx, split_size, num_heads, dtype=torch.float32,
                           qk_scale=None):
    """LePEAttention.forward -- vision_transformers/cswin.py:101-127 (im2cswin :78-84, get_lepe :86-99).

    qkv: (3, B, L, C').  Per (image, stripe window, head): out = softmax((q*scale) k^T) v + LePE, where
    LePE is the depth-wise 3x3 conv of v laid out as the (C', H_sp, W_sp) image of THAT window (zero
    padding at the window border, weight get_v.weight[c,0], bias get_v.bias[c]).  Output (B, L, C').
    """
    qkv = _t(qkv, dtype)
    wv, bv = _t(get_v_w, dtype), _t(get_v_b, dtype)
    _, B, L, C = qkv.shape
    H_sp, W_sp = _stripe_shape(reso, idx, split_size)
    T = H_sp * W_sp
    d = C // num_heads
    scale = qk_scale or d ** -0.5
    tab = window_token_index(reso, H_sp, W_sp)                       # (nWin, T)
    nwin = tab.shape[0]
    q = qkv[0][:, tab]                                               # (B, nWin, T, C)
    k = qkv[1][:, tab]
    v = qkv[2][:, tab]

    # LePE: depth-wise 3x3 on the window image, channel-last arithmetic with explicit zero halo.
    vimg = v.reshape(B, nwin, H_sp, W_sp, C)
    halo = torch.zeros(B, nwin, H_sp + 2, W_sp + 2, C, dtype=dtype)
    halo[:, :, 1:1 + H_sp, 1:1 + W_sp] = vimg
    lepe = torch.zeros(B, nwin, H_sp, W_sp, C, dtype=dtype) + bv
    for dy in range(3):
        for dx in range(3):
            lepe = lepe + halo[:, :, dy:dy + H_sp, dx:dx + W_sp] * wv[:, 0, dy, dx]
    lepe = lepe.reshape(B, nwin, T, C)

    def heads(z):                                                    # (B,nWin,T,C) -> (B,nWin,h,T,d)
        return z.reshape(B, nwin, T, num_heads, d).permute(0, 1, 3, 2, 4)

    o = sdpa_core(heads(q), heads(k), heads(v), scale, pre_scale=True)     # (B,nWin,h,T,d)
    o = o.permute(0, 1, 3, 2, 4).reshape(B, nwin, T, C) + lepe
    out = torch.empty(B, L, C, dtype=dtype)
    out[:, tab.reshape(-1)] = o.reshape(B, nwin * T, C)              # windows2img scatter
    return out


def cswin_block_forward(x, p, reso, num_heads, split_size, last_stage=False, dtype=torch.float32):
    """CSWinBlock.forward -- vision_transformers/cswin.py:176-197 (ctor :132-174).

    u = LN1(x); qkv = u Wqkv^T + b viewed (B,L,3,C) -> (3,B,L,C); branch 0 takes channels [:C/2] with
    vertical stripes (idx 0), branch 1 takes [C/2:] with horizontal stripes (idx 1), h/2 heads each;
    last stage (or reso == split_size, :146-147) uses one full-window branch.  proj, residual, then the
    single-GELU Mlp with residual.  proj_drop is declared but never applied (:153 vs :193).
    """
    x = _t(x, dtype)
    B, L, C = x.shape
    if reso == split_size:
        last_stage = True
    u = layernorm(x, _t(p["norm1.weight"], dtype), _t(p["norm1.bias"], dtype))
    qkv = linear(u, _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    qkv = qkv.reshape(B, L, 3, C).permute(2, 0, 1, 3)                # (3,B,L,C)
    if last_stage:
        att = lepe_attention_forward(qkv, p["attns.0.get_v.weight"], p["attns.0.get_v.bias"], reso, -1,
                                     split_size, num_heads, dtype)
    else:
        half = C // 2
        a0 = lepe_attention_forward(qkv[..., :half], p["attns.0.get_v.weight"], p["attns.0.get_v.bias"],
                                    reso, 0, split_size, num_heads // 2, dtype)
        a1 = lepe_attention_forward(qkv[..., half:], p["attns.1.get_v.weight"], p["attns.1.get_v.bias"],
                                    reso, 1, split_size, num_heads // 2, dtype)
        att = torch.cat([a0, a1], dim=2)
    x = x + linear(att, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))
    u = layernorm(x, _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
    m = _sub(p, "mlp.")
    h = gelu(linear(u, _t(m["fc1.weight"], dtype), _t(m["fc1.bias"], dtype)))
    return x + linear(h, _t(m["fc2.weight"], dtype), _t(m["fc2.bias"], dtype))


def cswin_forward(img, p, embed_dim=64, depth=(1, 2, 21, 1), split_size=(1, 2, 7, 7), num_heads=(2, 4, 8, 16), dtype=torch.float32):
    """CSWinTransformer.forward -- vision_transformers/cswin.py:324-346 (ctor :238-298; tiny-224 factory :360-363).

    stem = Conv2d(3, C, 7, stride 4, pad 2) -> tokens (b, h*w, c) -> LayerNorm (:247-251); per stage the CSWinBlocks; between
    stages Merge_Block = tokens -> NCHW -> Conv2d(C, 2C, 3, stride 2, pad 1) -> tokens -> LayerNorm (:224-233); final LayerNorm,
    mean over tokens, head.  Convolutions are the ATen conv2d the reference's nn.Conv2d runs.
    """
    x = _t(img, dtype)
    B, _, H, _ = x.shape
    x = TF.conv2d(x, _t(p["stage1_conv_embed.0.weight"], dtype), _t(p["stage1_conv_embed.0.bias"], dtype), stride=4, padding=2)
    reso = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)
    x = layernorm(x, _t(p["stage1_conv_embed.2.weight"], dtype), _t(p["stage1_conv_embed.2.bias"], dtype))
    for si in range(4):
        if si > 0:
            m = _sub(p, f"merge{si}.")
            C = x.shape[-1]
            grid = x.transpose(1, 2).reshape(B, C, reso, reso)
            grid = TF.conv2d(grid, _t(m["conv.weight"], dtype), _t(m["conv.bias"], dtype), stride=2, padding=1)
            reso = grid.shape[-1]
            x = layernorm(grid.flatten(2).transpose(1, 2), _t(m["norm.weight"], dtype), _t(m["norm.bias"], dtype))
        for bi in range(depth[si]):
            x = cswin_block_forward(x, _sub(p, f"stage{si + 1}.{bi}."), reso, num_heads[si], split_size[si], last_stage=(si == 3),
                                    dtype=dtype)
    x = layernorm(x, _t(p["norm.weight"], dtype), _t(p["norm.bias"], dtype))
    return linear(x.mean(dim=1), _t(p["head.weight"], dtype), _t(p["head.bias"], dtype))
