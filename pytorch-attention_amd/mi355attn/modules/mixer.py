"""Drop-in MLP-Mixer layer (reference: mlps/mlp_mixer.py:16-50), forward routed to libmi355attn.

Token mixing is a LEFT multiplication of each image's (N x C) matrix, so the reference's two transposes
(mlp_mixer.py:47) never materialise: the batched GEMM reads the activation as its K-major operand.
"""
from torch import nn

from .. import functional as F
from .vit import _fast, _mlp16


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.precision = None

    def forward(self, x, resid=None):
        if _fast(self.precision, self.fc1, self.fc2):
            return _mlp16(x, self.fc1, self.fc2, self.precision, second_gelu=False, resid=resid)
        h = F.linear(x, self.fc1.weight, self.fc1.bias, act=F.ACT_GELU, precision=self.precision)
        return F.linear(h, self.fc2.weight, self.fc2.bias, resid=resid, precision=self.precision)


class MixerLayer(nn.Module):
    def __init__(self, embedding_dim, sequence_len, mlp_ratio=[0.5, 4], drop=0, precision=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.token_mlp = Mlp(sequence_len, int(embedding_dim * mlp_ratio[0]), drop=drop)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.channel_mlp = Mlp(embedding_dim, int(embedding_dim * mlp_ratio[1]), drop=drop)
        self.precision = precision
        self.channel_mlp.precision = precision

    def forward(self, x):
        t = self.token_mlp
        p = F._prec(self.precision)
        T, N = t.fc1.weight.shape
        if x.dim() == 3 and t.fc1.bias is not None and t.fc2.bias is not None and F.mixer_token_ok(N, T, x.shape[-1], self.precision):
            # the whole half in one kernel: neither LN(x)^T nor the hidden tensor exists in HBM (csrc/mixer_fused.hip)
            x = F.mixer_token_mlp(x.contiguous(), self.norm1, t.fc1, t.fc2, p)
        elif p in (F.PREC_FP16, F.PREC_BF16) and T % 64 == 0 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024:
            # channel-major on the 16-bit GEMM engine: LN(x)^T (B,C,NP) -> gelu(. W1^T + b1) (B,C,T) -> (. W2^T + b2)^T + x
            NP = -(-N // 64) * 64
            ut = F.layernorm16_t(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, NP, p)
            ht = F.linear16(ut, F.weight16_padk(t.fc1.weight, NP, p), t.fc1.bias, act=F.ACT_GELU, out16=True, precision=p)
            x = F.linear16_tr(ht, F.weight16(t.fc2.weight, p), t.fc2.bias, x, p)
        else:
            u = F.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            hid = F.token_mix(t.fc1.weight, u, t.fc1.bias, act=F.ACT_GELU, precision=self.precision)     # (B,T,C)
            x = F.token_mix(t.fc2.weight, hid, t.fc2.bias, resid=x, precision=self.precision)            # (B,N,C)
        c = self.channel_mlp
        if _fast(self.precision, c.fc1, c.fc2):
            u = F.layernorm16(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, F._prec(self.precision))
        else:
            u = F.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return c(u, resid=x)


class PatchEmbedding(nn.Module):
    """mlp_mixer.py:52-63 (the argument really is spelled `in_chanels` in the reference)."""

    def __init__(self, image_size, patch_size, in_chanels=3, embedding_dim=768):
        super().__init__()
        assert image_size % patch_size == 0
        self.patch_size = patch_size
        self.num_patches = (image_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chanels, embedding_dim, kernel_size=patch_size, stride=patch_size)
        self.precision = None

    def forward(self, x):
        return F.patch_embed(x, self.proj.weight, self.proj.bias, None, None, self.patch_size, self.precision)


class MLP_Mixer(nn.Module):
    """Full MLP-Mixer (mlp_mixer.py:65-79): patch GEMM -> `depth` MixerLayers -> token mean -> head."""

    def __init__(self, dim=512, depth=12, image_size=224, patch_size=16, in_channels=3, drop=0, num_classes=1000, precision=None):
        super().__init__()
        self.patch_embedding = PatchEmbedding(image_size, patch_size, in_channels, dim)
        self.patch_embedding.precision = precision
        self.blocks = nn.Sequential(*[MixerLayer(dim, self.patch_embedding.num_patches, drop=drop, precision=precision)
                                      for _ in range(depth)])
        self.head = nn.Linear(dim, num_classes)
        self.precision = precision

    def forward(self, x):
        x = self.patch_embedding(x)
        for blk in self.blocks:
            x = blk(x)
        return F.linear(F.token_mean(x), self.head.weight, self.head.bias, precision=self.precision)
