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


# ---- the same block in the reference's ATen op sequence (bench.py's CPU leg) ----------------------------------------------------------
def _windows(t, reso, H_sp, W_sp):
    """(B, L, C) token-major image -> (B * nWin, T, C): the partition of img2windows (cswin.py:199-206) applied to token rows (the
    reference goes through a (B, C, H, W) view first, :80-83; the row order inside a window and the window order are the same)."""
    B, L, C = t.shape
    return t.reshape(B, reso // H_sp, H_sp, reso // W_sp, W_sp, C).transpose(2, 3).reshape(-1, H_sp * W_sp, C)


def _unwindows(w, B, reso, H_sp, W_sp):
    """inverse of _windows: windows2img (cswin.py:208-216) in token-major form."""
    C = w.shape[-1]
    return w.reshape(B, reso // H_sp, reso // W_sp, H_sp, W_sp, C).transpose(2, 3).reshape(B, reso * reso, C)


def lepe_attention_forward_aten(qkv, get_v_w, get_v_b, reso, idx, split_size, num_heads):
    """LePEAttention.forward (cswin.py:101-127) with the operator sequence the reference executes on the CPU: strided window views +
    one copy per operand, batched matmul, softmax, batched matmul, a grouped conv2d for LePE (get_lepe :86-99) -- instead of the
    index tables of lepe_attention_forward above, which are easy to audit but cost ~2x on a CPU.  Same arithmetic, fp32."""
    import torch.nn.functional as TF
    _, B, L, C = qkv.shape
    H_sp, W_sp = _stripe_shape(reso, idx, split_size)
    T, d = H_sp * W_sp, C // num_heads
    scale = d ** -0.5

    def heads(t):                                                    # (B*nWin, T, C) -> (B*nWin, h, T, d)
        return t.reshape(-1, T, num_heads, d).transpose(1, 2)

    qw, kw, vw = (_windows(qkv[i], reso, H_sp, W_sp) for i in range(3))
    vimg = vw.transpose(1, 2).reshape(-1, C, H_sp, W_sp)             # the window as a (C, H_sp, W_sp) image
    lepe = TF.conv2d(vimg, get_v_w, get_v_b, stride=1, padding=1, groups=C)
    lepe = lepe.reshape(-1, num_heads, d, T).transpose(2, 3)         # (B*nWin, h, T, d)
    attn = torch.softmax((heads(qw) * scale) @ heads(kw).transpose(-2, -1), dim=-1)
    out = attn @ heads(vw) + lepe
    return _unwindows(out.transpose(1, 2).reshape(-1, T, C), B, reso, H_sp, W_sp)


def cswin_block_forward_aten(x, p, reso, num_heads, split_size, last_stage=False):
    """CSWinBlock.forward (cswin.py:176-197) on torch's fused CPU operators (layer_norm, linear, gelu) and
    lepe_attention_forward_aten: what bench.py times as the CPU baseline of the CSWin blocks, so that the stand-in does not
    understate the reference (the index-table form above ran at half the reference's speed at stage 1).  Checked against
    cswin_block_forward in tests/test_oracle_golden.py."""
    import torch.nn.functional as TF
    B, L, C = x.shape
    if reso == split_size:
        last_stage = True
    u = TF.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"])
    qkv = TF.linear(u, p["qkv.weight"], p.get("qkv.bias")).reshape(B, L, 3, C).permute(2, 0, 1, 3)
    if last_stage:
        att = lepe_attention_forward_aten(qkv, p["attns.0.get_v.weight"], p["attns.0.get_v.bias"], reso, -1, split_size, num_heads)
    else:
        half = C // 2
        att = torch.cat([lepe_attention_forward_aten(qkv[..., :half], p["attns.0.get_v.weight"], p["attns.0.get_v.bias"], reso, 0, split_size, num_heads // 2),
                         lepe_attention_forward_aten(qkv[..., half:], p["attns.1.get_v.weight"], p["attns.1.get_v.bias"], reso, 1, split_size, num_heads // 2)], dim=2)
    x = x + TF.linear(att, p["proj.weight"], p["proj.bias"])
    u = TF.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"])
    return x + TF.linear(TF.gelu(TF.linear(u, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
