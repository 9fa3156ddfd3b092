"""TEST INFRASTRUCTURE -- CPU restatement (torch, fp32 / fp64) of the remaining members of the reference's attention zoo
(SURVEY 8 f2): GCModule, CoordinateAttention, TripletAttention, BAM, SKLayer, PAM / CAM.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product path never does.

Each function takes the reference module's state_dict (`sd`, keys as the reference names them) and follows the cited forward line
by line, BatchNorm in eval mode (running statistics), written with explicit formulas where the reference leans on nn layers so
that the algorithm is visible.  Pinned against outputs of the real reference (tests/golden/, tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as TF


def _t(v, dtype):
    return v.detach().to("cpu", dtype)


def _bn(z, sd, prefix, dtype, eps=1e-5):
    """Eval-mode BatchNorm over dim 1 (nn.BatchNorm1d / 2d): (z - running_mean) / sqrt(running_var + eps) * weight + bias."""
    shape = [1, -1] + [1] * (z.dim() - 2)
    g = lambda k: _t(sd[prefix + "." + k], dtype).reshape(shape)
    return (z - g("running_mean")) / torch.sqrt(g("running_var") + eps) * g("weight") + g("bias")


def gc_forward(x, sd, dtype=torch.float32):
    """GCModule.forward -- attention_mechanisms/gc_module.py:30-43.  context_modeling (:30-38): a 1x1 conv gives one logit per
    position; the declared softmax (:22) is NEVER applied, so ctx[b,c] = sum_hw x[b,c,hw] * logit[b,hw].  transform (:23-28):
    conv1x1 -> LayerNorm([Cr,1,1]) -> ReLU -> conv1x1.  Fusion (:43): x + y broadcast over positions."""
    x = _t(x, dtype)
    b, c, h, w = x.shape
    logit = TF.conv2d(x, _t(sd["conv.weight"], dtype), _t(sd["conv.bias"], dtype)).reshape(b, 1, h * w)
    ctx = torch.matmul(x.reshape(b, c, h * w), logit.transpose(1, 2))                     # (b, c, 1)
    w1, b1 = _t(sd["transform.0.weight"], dtype), _t(sd["transform.0.bias"], dtype)
    z = w1.reshape(w1.shape[0], c) @ ctx + b1[None, :, None]                              # (b, Cr, 1)
    mu = z.mean(dim=1, keepdim=True)
    var = ((z - mu) ** 2).mean(dim=1, keepdim=True)
    z = (z - mu) / torch.sqrt(var + 1e-5) * _t(sd["transform.1.weight"], dtype).reshape(1, -1, 1) \
        + _t(sd["transform.1.bias"], dtype).reshape(1, -1, 1)
    z = torch.relu(z)
    w2, b2 = _t(sd["transform.3.weight"], dtype), _t(sd["transform.3.bias"], dtype)
    y = w2.reshape(c, -1) @ z + b2[None, :, None]
    return x + y.reshape(b, c, 1, 1)


def coordatt_forward(x, sd, dtype=torch.float32):
    """CoordinateAttention.forward -- attention_mechanisms/coordatten.py:30-44: row means and column means concatenated along the
    position axis, shared conv1 + bn1 + ReLU, split, conv_h / conv_w; out = x * a_h * a_w (the reference applies NO sigmoid)."""
    x = _t(x, dtype)
    b, c, h, w = x.shape
    x_h = x.mean(dim=3, keepdim=True)                                                      # (b,c,h,1)
    x_w = x.mean(dim=2, keepdim=True).transpose(-1, -2)                                    # (b,c,w,1)
    y = torch.cat([x_h, x_w], dim=2)
    y = TF.conv2d(y, _t(sd["conv1.weight"], dtype), _t(sd["conv1.bias"], dtype))
    y = torch.relu(_bn(y, sd, "bn1", dtype))
    y_h, y_w = torch.split(y, [h, w], dim=2)
    a_h = TF.conv2d(y_h, _t(sd["conv_h.weight"], dtype), _t(sd["conv_h.bias"], dtype))    # (b,c,h,1)
    a_w = TF.conv2d(y_w.transpose(-1, -2), _t(sd["conv_w.weight"], dtype), _t(sd["conv_w.bias"], dtype))   # (b,c,1,w)
    return x * a_h * a_w


def _attention_gate(x, sd, prefix, dtype):
    """AttentionGate.forward -- triplet_attention.py:45-49 with ZPool :32-36 and BasicConv2d :27-31 (conv -> bn -> ReLU) -> sigmoid."""
    z = torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)
    wt = _t(sd[prefix + ".conv.conv.weight"], dtype)
    z = TF.conv2d(z, wt, _t(sd[prefix + ".conv.conv.bias"], dtype), padding=(wt.shape[-1] - 1) // 2)
    z = torch.relu(_bn(z, sd, prefix + ".conv.bn", dtype))
    return x * torch.sigmoid(z)


def triplet_forward(x, sd, dtype=torch.float32):
    """TripletAttention.forward -- triplet_attention.py:58-63: the gate applied to three rotations of x, averaged."""
    x = _t(x, dtype)
    x_ch = _attention_gate(x.permute(0, 3, 1, 2), sd, "ch", dtype).permute(0, 2, 3, 1)
    x_cw = _attention_gate(x.permute(0, 2, 1, 3), sd, "cw", dtype).permute(0, 2, 1, 3)
    x_hw = _attention_gate(x, sd, "hw", dtype)
    return 1 / 3 * (x_ch + x_cw + x_hw)


def bam_forward(x, sd, dilation=4, dtype=torch.float32):
    """BAM.forward -- attention_mechanisms/bam.py:63-71 with ChannelGate :27-33 (avgpool -> Linear -> ReLU -> Linear -> BatchNorm1d)
    and SpatialGate :52-59 (1x1 conv -> two dilated 3x3 conv + BN + ReLU -> 1x1 conv to one plane -> BN): x + x * sigmoid(cg + sg)."""
    x = _t(x, dtype)
    g = lambda k: _t(sd[k], dtype)
    s = x.mean(dim=(2, 3))
    z = torch.relu(s @ g("channel_attn.mlp.0.weight").t() + g("channel_attn.mlp.0.bias"))
    z = z @ g("channel_attn.mlp.2.weight").t() + g("channel_attn.mlp.2.bias")
    cg = _bn(z, sd, "channel_attn.bn", dtype)[:, :, None, None]
    y = TF.conv2d(x, g("spatial_attn.conv1.weight"), g("spatial_attn.conv1.bias"))
    for conv, bn in (("0", "1"), ("3", "4")):
        y = TF.conv2d(y, g(f"spatial_attn.conv2.{conv}.weight"), g(f"spatial_attn.conv2.{conv}.bias"), padding=dilation, dilation=dilation)
        y = torch.relu(_bn(y, sd, f"spatial_attn.conv2.{bn}", dtype))
    y = TF.conv2d(y, g("spatial_attn.conv3.weight"), g("spatial_attn.conv3.bias"))
    sg = _bn(y, sd, "spatial_attn.bn", dtype)
    return x + x * torch.sigmoid(cg + sg)


def sk_forward(x, sd, groups=32, dtype=torch.float32):
    """SKLayer.forward -- attention_mechanisms/sk_module.py:41-56: two grouped 3x3 branches (dilation 1 and 2) + BN + ReLU, fused by
    a softmax over the two branches computed from the pooled sum."""
    x = _t(x, dtype)
    g = lambda k: _t(sd[k], dtype)
    u1 = torch.relu(_bn(TF.conv2d(x, g("split_3x3.0.weight"), g("split_3x3.0.bias"), padding=1, groups=groups), sd, "split_3x3.1", dtype))
    u2 = torch.relu(_bn(TF.conv2d(x, g("split_5x5.0.weight"), g("split_5x5.0.bias"), padding=2, dilation=2, groups=groups), sd,
                        "split_5x5.1", dtype))
    s = (u1 + u2).mean(dim=(2, 3))
    z = torch.relu(_bn(s @ g("fc.0.weight").t() + g("fc.0.bias"), sd, "fc.1", dtype))
    a = z @ g("fc1.weight").t() + g("fc1.bias")
    b = z @ g("fc2.weight").t() + g("fc2.bias")
    att = torch.softmax(torch.stack([a, b], dim=1), dim=1)
    return u1 * att[:, 0, :, None, None] + u2 * att[:, 1, :, None, None]


def pam_forward(x, sd, dtype=torch.float32):
    """PAM.forward -- attention_mechanisms/dual_attention.py:20-28: position attention over the hw positions, unscaled logits."""
    x = _t(x, dtype)
    n, c, h, w = x.shape
    g = lambda k: _t(sd[k], dtype)
    B = TF.conv2d(x, g("b.weight"), g("b.bias")).flatten(2).transpose(1, 2)
    C = TF.conv2d(x, g("c.weight"), g("c.bias")).flatten(2)
    D = TF.conv2d(x, g("d.weight"), g("d.bias")).flatten(2).transpose(1, 2)
    attn = torch.softmax(B @ C, dim=-1)
    y = (attn @ D).transpose(1, 2).reshape(n, c, h, w)
    return g("alpha") * y + x


def cam_forward(x, sd, dtype=torch.float32):
    """CAM.forward -- attention_mechanisms/dual_attention.py:35-42: channel attention from the C x C Gram matrix, unscaled logits."""
    x = _t(x, dtype)
    b, c, h, w = x.shape
    x_ = x.flatten(2)
    attn = torch.softmax(x_ @ x_.transpose(1, 2), dim=-1)
    return _t(sd["beta"], dtype) * (attn @ x_).reshape(b, c, h, w) + x
