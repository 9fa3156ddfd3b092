"""Oracle (test infrastructure): ATen-OPERATOR-SEQUENCE restatements of the reference forwards that bench.py times on the host.

The functions of oracle/transformer.py, oracle/xcit.py and oracle/chan_attn.py restate the MATH (explicit mean / variance passes, an
erf formula, a Python loop over heads, index tables): right for judging parity, several times slower on a CPU than what the reference
itself runs, because the reference's nn.Modules call the FUSED ATen kernels (`layer_norm`, `linear` = addmm, `gelu`, `softmax`, batched
`matmul` over a 4-D view).  bench.py's `cpu_baseline` leg is "the reference's CPU path timed beside the GPU", so it must run the same
operators in the same order; these functions do, functionally (weights come in as a state_dict, nothing is an nn.Module), each citing
the reference lines it follows.  /root/reference does not exist on the GPU box, hence a restatement and `kind: "port"`.

tests/test_oracle_golden.py checks every function here against the math restatements, the golden records and -- in the build
container -- the live reference modules (<= 2e-6 relative: same operators, same order, same bits up to threading).
"""
import torch
import torch.nn.functional as TF


def _g(p, k):
    return p[k] if k in p else None


def vit_attention_aten(x, p, num_heads):
    """Attention.forward -- vision_transformers/ViT.py:79-89: qkv Linear -> reshape (B,N,3,h,d) -> permute(2,0,3,1,4) -> unbind ->
    (q @ k^T) * scale -> softmax(-1) -> attn @ v -> transpose(1,2).reshape(B,N,C) -> proj Linear."""
    B, N, C = x.shape
    d = C // num_heads
    qkv = TF.linear(x, p["qkv.weight"], _g(p, "qkv.bias")).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return TF.linear(y, p["proj.weight"], p["proj.bias"])


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def vit_encoder_aten(x, p, num_heads):
    """TransformerEncoder.forward -- ViT.py:116-119; Mlp.forward :58-65 (GELU after fc1 AND after fc2)."""
    C = x.shape[-1]
    u = TF.layer_norm(x, (C,), p["layernorm1.weight"], p["layernorm1.bias"])
    x = x + vit_attention_aten(u, _sub(p, "attn."), num_heads)
    u = TF.layer_norm(x, (C,), p["layernorm2.weight"], p["layernorm2.bias"])
    h = TF.gelu(TF.linear(u, p["mlp.fc1.weight"], p["mlp.fc1.bias"]))
    return x + TF.gelu(TF.linear(h, p["mlp.fc2.weight"], p["mlp.fc2.bias"]))


def vit_aten(img, p, num_heads, depths):
    """VisionTransformer.forward at the native resolution -- ViT.py:180-192: conv patch embedding (:101-105) -> flatten(2).transpose ->
    cat([patches, cls]) (cls LAST) -> + position_embedding -> blocks -> head(x[:, 0])."""
    x = TF.conv2d(img, p["patch_embedding.proj.weight"], p["patch_embedding.proj.bias"], stride=p["patch_embedding.proj.weight"].shape[-1])
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([x, p["cls_token"].expand(x.shape[0], -1, -1)], dim=1)
    x = x + p["position_embedding"]
    for i in range(depths):
        x = vit_encoder_aten(x, _sub(p, "blocks.%d." % i), num_heads)
    return TF.linear(x[:, 0], p["head.weight"], p["head.bias"])


def xca_aten(x, p, num_heads):
    """XCA.forward -- vision_transformers/xcit.py:245-265: qkv Linear -> (3,B,h,N,d) -> transpose to (B,h,d,N) -> F.normalize(dim=-1) on
    q and k -> (q @ k^T) * temperature -> softmax -> attn @ v -> permute(0,3,1,2).reshape(B,N,C) -> proj."""
    B, N, C = x.shape
    d = C // num_heads
    qkv = TF.linear(x, p["qkv.weight"], _g(p, "qkv.bias")).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q, k, v = q.transpose(-2, -1), k.transpose(-2, -1), v.transpose(-2, -1)
    q = TF.normalize(q, dim=-1)
    k = TF.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * p["temperature"]
    attn = attn.softmax(dim=-1)
    y = (attn @ v).permute(0, 3, 1, 2).reshape(B, N, C)
    return TF.linear(y, p["proj.weight"], p["proj.bias"])


def lpi_aten(x, p, H, W):
    """LPI.forward -- xcit.py:149-157: tokens -> (B,C,H,W) -> depth-wise conv3x3 -> GELU -> BatchNorm2d (eval: running statistics) ->
    depth-wise conv3x3 -> tokens."""
    B, N, C = x.shape
    x = x.transpose(1, 2).reshape(B, C, H, W)
    x = TF.conv2d(x, p["conv1.weight"], p["conv1.bias"], padding=1, groups=C)
    x = TF.gelu(x)
    x = TF.batch_norm(x, p["bn.running_mean"], p["bn.running_var"], p["bn.weight"], p["bn.bias"], False, 0.1, 1e-5)
    x = TF.conv2d(x, p["conv2.weight"], p["conv2.bias"], padding=1, groups=C)
    return x.reshape(B, C, N).transpose(1, 2)


def xca_block_aten(x, p, num_heads, H, W):
    """XCABlock.forward -- xcit.py:290-294: x + g1 * XCA(LN1 x); x + g3 * LPI(LN3 x); x + g2 * Mlp(LN2 x) (Mlp :32-38: one GELU)."""
    C = x.shape[-1]
    x = x + p["gamma1"] * xca_aten(TF.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"]), _sub(p, "attn."), num_heads)
    x = x + p["gamma3"] * lpi_aten(TF.layer_norm(x, (C,), p["norm3.weight"], p["norm3.bias"]), _sub(p, "local_mp."), H, W)
    u = TF.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"])
    h = TF.gelu(TF.linear(u, p["mlp.fc1.weight"], p["mlp.fc1.bias"]))
    return x + p["gamma2"] * TF.linear(h, p["mlp.fc2.weight"], p["mlp.fc2.bias"])


def mixer_layer_aten(x, p):
    """MixerLayer.forward -- mlps/mlp_mixer.py:45-50: x + token_mlp(norm1(x).transpose(1,2)).transpose(1,2); x + channel_mlp(norm2(x));
    Mlp :27-33 (one GELU)."""
    C = x.shape[-1]
    u = TF.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"]).transpose(1, 2)
    h = TF.gelu(TF.linear(u, p["token_mlp.fc1.weight"], p["token_mlp.fc1.bias"]))
    x = x + TF.linear(h, p["token_mlp.fc2.weight"], p["token_mlp.fc2.bias"]).transpose(1, 2)
    u = TF.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"])
    h = TF.gelu(TF.linear(u, p["channel_mlp.fc1.weight"], p["channel_mlp.fc1.bias"]))
    return x + TF.linear(h, p["channel_mlp.fc2.weight"], p["channel_mlp.fc2.bias"])


def double_attention_aten(x, p):
    """DoubleAttention.forward -- attention_mechanisms/double_attention.py:32-48: three 1x1 convs -> softmax over HW of B, softmax over
    c_n of V -> bmm(A, maps^T) -> matmul(G, vectors) -> 1x1 conv."""
    b, c, h, w = x.shape
    A = TF.conv2d(x, p["convA.weight"], p["convA.bias"])
    Bm = TF.conv2d(x, p["convB.weight"], p["convB.bias"])
    V = TF.conv2d(x, p["convV.weight"], p["convV.bias"])
    cm, cn = A.shape[1], Bm.shape[1]
    maps = TF.softmax(Bm.view(b, cn, h * w), dim=-1)
    G = torch.bmm(A.view(b, cm, h * w), maps.permute(0, 2, 1))
    vec = TF.softmax(V.view(b, cn, h * w), dim=1)
    Z = G.matmul(vec).view(b, cm, h, w)
    return TF.conv2d(Z, p["proj.weight"], p["proj.bias"])


def se_aten(x, p):
    """SELayer.forward -- attention_mechanisms/se_module.py:29-33: AdaptiveAvgPool2d(1) -> Linear -> ReLU -> Linear -> Sigmoid -> x * y."""
    b, c = x.shape[:2]
    y = TF.adaptive_avg_pool2d(x, 1).view(b, c)
    y = torch.sigmoid(TF.linear(torch.relu(TF.linear(y, p["fc.0.weight"])), p["fc.2.weight"]))
    return x * y.view(b, c, 1, 1).expand_as(x)


def eca_aten(x, p):
    """ECALayer.forward -- attention_mechanisms/eca.py:26-30: avg pool -> Conv1d over the channel axis -> sigmoid -> x * y."""
    y = TF.adaptive_avg_pool2d(x, 1)
    k = p["conv.weight"].shape[-1]
    y = TF.conv1d(y.squeeze(-1).transpose(-1, -2), p["conv.weight"], padding=(k - 1) // 2).transpose(-1, -2).unsqueeze(-1)
    return x * torch.sigmoid(y).expand_as(x)


def cbam_aten(x, p):
    """CBAM.forward -- attention_mechanisms/cbam.py:56-59; ChannelAttention :31-35 (shared 1x1-conv MLP on the avg- and max-pooled
    maps, summed, sigmoid, multiply); SpatialAttention :43-48 (cat[mean_c, max_c] -> conv kxk -> sigmoid, multiply)."""
    def fc(v):
        return TF.conv2d(torch.relu(TF.conv2d(v, p["ca.fc.0.weight"])), p["ca.fc.2.weight"])
    g = torch.sigmoid(fc(TF.adaptive_avg_pool2d(x, 1)) + fc(TF.adaptive_max_pool2d(x, 1)))
    x = x * g
    s = torch.cat([torch.mean(x, dim=1, keepdim=True), torch.max(x, dim=1, keepdim=True)[0]], dim=1)
    ks = p["sa.conv.weight"].shape[-1]
    return x * torch.sigmoid(TF.conv2d(s, p["sa.conv.weight"], padding=ks // 2))
