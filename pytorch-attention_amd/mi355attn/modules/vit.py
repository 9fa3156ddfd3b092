"""Drop-in ViT modules (reference: vision_transformers/ViT.py), forward composed from libmi355attn ops.

  Attention           ViT.py:67-89      qkv GEMM -> fused QK^T-softmax-PV core -> proj GEMM(+bias)
  Mlp                 ViT.py:47-65      GELU after fc1 AND after fc2 (reference behaviour, kept)
  TransformerEncoder  ViT.py:107-119    pre-LN block, residual adds fused into the GEMM epilogues
  PatchEmbedding      ViT.py:91-105     per-patch GEMM, im2col folded into the A-operand load
  VisionTransformer   ViT.py:121-192    cls token appended LAST, logits from token 0, no final LayerNorm

Sub-modules hold parameters only (same creation order as the reference => same init stream under a seed).
``precision`` (None = package default) selects the MFMA operand format for every GEMM of the module.
"""
import torch
from torch import nn

from .. import functional as F


def _bias(lin):
    return lin.bias if lin.bias is not None else None


def _fast(precision, *linears):
    """True when the 16-bit dataflow applies: a 16-bit operand mode and every GEMM inside the fast kernel's envelope."""
    return F._prec(precision) != F.PREC_STRICT and all(F.fast_gemm_ok(l.in_features, l.out_features) for l in linears)


def _mlp16(x, fc1, fc2, precision, second_gelu, gamma=None, resid=None):
    """fc1 -> GELU -> fc2 (-> GELU) with the hidden activation kept in the MFMA operand format; x is fp32 or 16-bit."""
    p = F._prec(precision)
    h16 = F.cast_linear16(x, F.weight16(fc1.weight, p), fc1.bias, act=F.ACT_GELU, precision=p)    # fp32 x: the cast rides in the GEMM where it can
    folded = F.weight16_scaled(fc2.weight, fc2.bias, gamma, p) if gamma is not None and not second_gelu else None
    if folded is not None:                                    # LayerScale folded into fc2 (XCiT: no activation behind fc2); None =
        w16, b = folded                                       # gamma * W would leave the fp16 normal range: gamma stays in the epilogue
        return F.linear16(h16, w16, b, resid=resid, precision=p)
    return F.linear16(h16, F.weight16(fc2.weight, p), fc2.bias, act=F.ACT_GELU if second_gelu else F.ACT_NONE, gamma=gamma,
                      resid=resid, precision=p)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.precision = None
        self.drop = drop                     # inference engine: dropout is the identity (eval semantics)

    def forward(self, x, resid=None):
        if _fast(self.precision, self.fc1, self.fc2):
            return _mlp16(x, self.fc1, self.fc2, self.precision, second_gelu=True, resid=resid)
        h = F.linear(x, self.fc1.weight, self.fc1.bias, act=F.ACT_GELU, precision=self.precision)
        return F.linear(h, self.fc2.weight, self.fc2.bias, act=F.ACT_GELU, resid=resid, precision=self.precision)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=4, qkv_bias=False, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.precision = precision
        self.attn_drop, self.proj_drop = attn_drop, proj_drop   # identity at inference

    def fast_ok(self, n_tokens=None):
        """16-bit dataflow applies: 16-bit operand mode, GEMM shapes in the fast envelope, a core kernel built for the head width
        (the K/V-resident kernel for d in {32, 64} and N <= 224, the streaming kernel for any N and d in F.SDPA_WIDTHS)."""
        d = self.qkv.in_features // self.num_heads
        return _fast(self.precision, self.qkv, self.proj) and d in F.SDPA_WIDTHS

    def _core(self, qkv, fast):
        """softmax(q k^T scale) v on the (B,N,3C) projection.  Short sequences of 32 / 64 wide heads stay in LDS (attn.hip); longer
        sequences (384 px input: N = 577) and wider heads (ViT.py:68 defaults to 4 heads: d = 192 at dim 768) stream K / V
        (sdpa_general.hip); any other width runs zero padded to the next built one."""
        B, N, C3 = qkv.shape
        C = C3 // 3
        d = C // self.num_heads
        if d in (32, 64) and N <= 224:
            return F.sdpa16(qkv, self.num_heads, self.scale, precision=self.precision) if fast else \
                F.sdpa(qkv, self.num_heads, self.scale, precision=self.precision)
        return F.sdpa_general(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.num_heads, self.scale, precision=self.precision)

    def forward(self, x, resid=None):
        d = self.qkv.in_features // self.num_heads
        if d not in F.SDPA_WIDTHS:                                                           # odd head width: padded projections
            from .mhsa import _fused_qkv_attention
            out, _, _ = _fused_qkv_attention(x if x.dtype == torch.float32 else x.float(), self.qkv, self.proj, self.num_heads, d,
                                             self.scale, self.precision)
            return out if resid is None else F.axpby(out, torch.empty_like(out), out.numel() // out.shape[-1], out.shape[-1],
                                                     out.shape[-1], out.shape[-1], u=resid, ldu=out.shape[-1])
        if self.fast_ok():
            # q / k / v / ctx stay 16-bit in HBM; the whole block is ONE C call (mi355_mhsa_fwd: cast -> qkv GEMM -> core -> proj GEMM,
            # the same kernels the three-call composition of rounds 1-2 launched)
            p = F._prec(self.precision)
            return F.mhsa16(x, F.weight16(self.qkv.weight, p), _bias(self.qkv), F.weight16(self.proj.weight, p), self.proj.bias,
                            self.num_heads, self.scale, resid=resid, precision=p)
        qkv = F.linear(x, self.qkv.weight, _bias(self.qkv), precision=self.precision)      # (B,N,3C)
        ctx = self._core(qkv, False)                                                        # (B,N,C)
        return F.linear(ctx, self.proj.weight, self.proj.bias, resid=resid, precision=self.precision)


class PatchEmbedding(nn.Module):
    def __init__(self, image_size=224, patch_size=16, in_channels=3, embedding_dim=768):
        super().__init__()
        assert image_size % patch_size == 0
        self.patch_size = patch_size
        self.num_patches = (image_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_channels, embedding_dim, kernel_size=patch_size, stride=patch_size)
        self.precision = None

    def forward(self, x):
        return F.patch_embed(x, self.proj.weight, self.proj.bias, None, None, self.patch_size, self.precision)


class TransformerEncoder(nn.Module):
    def __init__(self, dim, num_heads=4, mlp_ratio=4, qkv_bias=False, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        self.attn = Attention(dim, num_heads, qkv_bias, attn_drop, proj_drop, precision)
        self.layernorm1 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.mlp.precision = precision
        self.layernorm2 = nn.LayerNorm(dim)

    def _norm(self, ln, x, fast):
        if fast:                     # LayerNorm writes the GEMM's operand format directly
            return F.layernorm16(x, ln.weight, ln.bias, ln.eps, F._prec(self.attn.precision))
        return F.layernorm(x, ln.weight, ln.bias, ln.eps)

    def forward(self, x):
        if self.fold_ok(x):
            return self.forward_folded(x)[0]
        return self.forward_plain(x)

    def forward_plain(self, x):
        """The default body: one LayerNorm launch in front of each half (the fold is opt-in: measured slower)."""
        fast = self.attn.fast_ok(x.shape[1]) and _fast(self.attn.precision, self.mlp.fc1, self.mlp.fc2)
        x = self.attn(self._norm(self.layernorm1, x, fast), resid=x)
        return self.mlp(self._norm(self.layernorm2, x, fast), resid=x)

    # ---- LayerNorm folded into the GEMMs around it (csrc/ln_fold.hip): no LayerNorm launch between proj and fc1, nor between fc2 and
    #      the next block's qkv; the first LayerNorm of a chain is one mi355_ln_center16_fwd ---------------------------------------
    def fold_ok(self, x):
        """The folded path applies: 16-bit dataflow, K/V-resident attention core, the producer kernel's shape envelope
        (rows % 128 == 0: B = 128, 256, ... at 197 tokens) and LayerNorm gains that keep gamma * W inside fp16."""
        if not F.ln_fold_enabled() or x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:   # the option first: off by default
            return False
        B, N, C = x.shape
        p = F._prec(self.attn.precision)
        d = C // self.attn.num_heads
        if not (self.attn.fast_ok(N) and _fast(p, self.mlp.fc1, self.mlp.fc2) and d in (32, 64) and N <= 224):
            return False
        if not (F.ln_fold_ok(B * N, C, C, C, p) and F.ln_fold_ok(B * N, C, C, self.mlp.fc1.out_features, p)):
            return False
        return (F.lnfold_weights(self.layernorm1, self.attn.qkv, p) is not None and
                F.lnfold_weights(self.layernorm2, self.mlp.fc1, p) is not None)

    def forward_folded(self, x, state=None, next_eps=None):
        """x (B,N,C) fp32 -> (y, LnState of the NEXT block's LayerNorm 1 or None).  `state`: the LnState of THIS block's LayerNorm 1
        when the previous block emitted it (else computed here from x); `next_eps`: eps of the next block's LayerNorm 1 when fc2
        should emit for it."""
        p = F._prec(self.attn.precision)
        ln1, ln2, at, mlp = self.layernorm1, self.layernorm2, self.attn, self.mlp
        wq, sq, bq = F.lnfold_weights(ln1, at.qkv, p)
        w1, s1, b1 = F.lnfold_weights(ln2, mlp.fc1, p)
        if state is None:
            state = F.ln_center16(x, ln1.eps, p)
        qkv = F.linear16_lnfold(state, wq, bq, sq, precision=p)                                  # LayerNorm 1 + qkv (+ bias)
        ctx = F.sdpa16(qkv, at.num_heads, at.scale, precision=p)
        x1, st = F.linear16_emit(ctx, F.weight16(at.proj.weight, p), at.proj.bias, x, state.cvec, ln2.eps, precision=p)   # proj + residual, emits LN2
        h = F.linear16_lnfold(st, w1, b1, s1, act=F.ACT_GELU, precision=p)                     # LayerNorm 2 + fc1 + GELU
        w2 = F.weight16(mlp.fc2.weight, p)
        if next_eps is None:
            return F.linear16(h, w2, mlp.fc2.bias, act=F.ACT_GELU, resid=x1, precision=p), None
        return F.linear16_emit(h, w2, mlp.fc2.bias, x1, st.cvec, next_eps, act=F.ACT_GELU, precision=p)   # fc2 + GELU + residual, emits the next LN1


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=16, in_channels=3, depths=12, num_heads=4, mlp_ratio=4,
                 embedding_dim=768, qkv_bias=False, attn_drop=0, proj_drop=0, global_pool="token",
                 num_classes=1000, precision=None):
        super().__init__()
        self.patch_embedding = PatchEmbedding(image_size, patch_size, in_channels, embedding_dim)
        self.patch_embedding.precision = precision
        self.global_pool = global_pool
        self.precision = precision
        n = self.patch_embedding.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embedding_dim))
        self.position_embedding = nn.Parameter(torch.zeros(1, n + 1, embedding_dim))
        self.blocks = nn.Sequential(*(TransformerEncoder(embedding_dim, num_heads, mlp_ratio, qkv_bias, attn_drop,
                                                         proj_drop, precision) for _ in range(depths)))
        self.head = nn.Linear(embedding_dim, num_classes)
        # same init stream as ViT.py:147-158: pos, cls, then every Linear (trunc-normal .02, zero bias) / LayerNorm
        nn.init.trunc_normal_(self.position_embedding, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(self._reset)

    @staticmethod
    def _reset(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    def forward(self, x):
        B, _, H, W = x.shape
        ps = self.patch_embedding.patch_size
        pos = F.vit_pos_table(self.position_embedding, ps, H, W)     # ViT.py:160-178: bicubic resize off the native grid (cached)
        tok = F.patch_embed(x, self.patch_embedding.proj.weight, self.patch_embedding.proj.bias,
                            self.cls_token.reshape(-1), pos, ps, self.precision)
        blocks, state = list(self.blocks), None
        # fold eligibility is decided ONCE per forward (the token tensor keeps its shape through the blocks): with the option off -- the
        # default -- no per-block option reads / envelope checks / cache look-ups happen at all
        elig = [blk.fold_ok(tok) for blk in blocks] if F.ln_fold_enabled() else None
        for i, blk in enumerate(blocks):
            if elig is not None and elig[i]:                      # LayerNorms folded into the GEMMs; the state travels block to block
                nxt_ok = i + 1 < len(blocks) and elig[i + 1]
                tok, state = blk.forward_folded(tok, state, blocks[i + 1].layernorm1.eps if nxt_ok else None)
            else:
                tok, state = blk(tok), None                       # module call (hooks intact); its own fold check is one option read
        if self.global_pool == "token":
            pooled = tok[:, 0]                                   # row-strided view, consumed in place by the GEMM
        elif self.global_pool == "avg":
            pooled = F.token_mean(tok, skip_first=1)
        else:
            pooled = tok                                         # ViT.py:187-190: any other value pools nothing, head on every token
        return F.linear(pooled, self.head.weight, self.head.bias, precision=self.precision)
