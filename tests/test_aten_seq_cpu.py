"""CPU tests: the ATen-operator-sequence restatements bench.py times on the host (oracle/aten_seq.py) agree with the math restatements
of the oracle, and -- in the build container -- with the LIVE reference modules they restate (same operators in the same order)."""
import importlib
import sys

import pytest
import torch

import oracle as O
from oracle import aten_seq as A
from conftest import REFERENCE, have_reference, rel_fro


def _mods():
    from mi355attn.modules import (CBAM, Attention, DoubleAttention, ECALayer, MixerLayer, SELayer, TransformerEncoder, VisionTransformer,
                                   XCA, XCABlock)
    return dict(SELayer=SELayer, CBAM=CBAM, ECALayer=ECALayer, DoubleAttention=DoubleAttention, Attention=Attention,
                TransformerEncoder=TransformerEncoder, VisionTransformer=VisionTransformer, XCA=XCA, XCABlock=XCABlock, MixerLayer=MixerLayer)


def _sd(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


CASES = [
    # name, ctor, input shape, aten function, math-oracle function
    ("se", lambda M: M["SELayer"](64), (3, 64, 14, 10), lambda x, p: A.se_aten(x, p),
     lambda x, p: O.se_forward(x, p["fc.0.weight"], p["fc.2.weight"])),
    ("eca", lambda M: M["ECALayer"](64), (3, 64, 14, 10), lambda x, p: A.eca_aten(x, p), lambda x, p: O.eca_forward(x, p["conv.weight"])),
    ("cbam", lambda M: M["CBAM"](64), (3, 64, 14, 10), lambda x, p: A.cbam_aten(x, p),
     lambda x, p: O.cbam_forward(x, p["ca.fc.0.weight"], p["ca.fc.2.weight"], p["sa.conv.weight"])),
    ("da", lambda M: M["DoubleAttention"](64, 32, 32), (2, 64, 12, 12), lambda x, p: A.double_attention_aten(x, p),
     lambda x, p: O.double_attention_forward(x, *[p[k] for k in ("convA.weight", "convA.bias", "convB.weight", "convB.bias",
                                                                  "convV.weight", "convV.bias", "proj.weight", "proj.bias")])),
    ("vit_attn", lambda M: M["Attention"](192, 6), (2, 50, 192), lambda x, p: A.vit_attention_aten(x, p, 6),
     lambda x, p: O.vit_attention_forward(x, p, 6)),
    ("vit_enc", lambda M: M["TransformerEncoder"](192, 6), (2, 50, 192), lambda x, p: A.vit_encoder_aten(x, p, 6),
     lambda x, p: O.vit_encoder_forward(x, p, 6)),
    ("vit", lambda M: M["VisionTransformer"](image_size=32, patch_size=8, depths=2, num_heads=4, embedding_dim=64, num_classes=10),
     (2, 3, 32, 32), lambda x, p: A.vit_aten(x, p, 4, 2), lambda x, p: O.vit_forward(x, p, 4, 2)),
    ("xca", lambda M: M["XCA"](96, 4, qkv_bias=True), (2, 36, 96), lambda x, p: A.xca_aten(x, p, 4), lambda x, p: O.xca_forward(x, p, 4)),
    ("xca_block", lambda M: M["XCABlock"](96, 4, qkv_bias=True, eta=1.0), (2, 36, 96), lambda x, p: A.xca_block_aten(x, p, 4, 6, 6),
     lambda x, p: O.xca_block_forward(x, p, 4, 6, 6)),
    ("mixer", lambda M: M["MixerLayer"](64, 49), (2, 49, 64), lambda x, p: A.mixer_layer_aten(x, p), lambda x, p: O.mixer_layer_forward(x, p)),
]


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_aten_sequence_equals_math_restatement(name):
    _, ctor, shape, f_aten, f_math = next(c for c in CASES if c[0] == name)
    torch.manual_seed(1234)
    m = ctor(_mods()).eval()
    for k, v in m.state_dict().items():                        # non-trivial LayerNorm / BatchNorm / LayerScale parameters
        if v.dtype.is_floating_point and v.ndim == 1:
            v.copy_(v + 0.1 * torch.randn_like(v))
        if k.endswith("running_var"):
            v.copy_(v.abs() + 0.5)
    torch.manual_seed(4321)
    x = torch.randn(*shape)
    p = _sd(m)
    with torch.no_grad():
        ya, ym = f_aten(x, p), f_math(x, p)
    assert ya.shape == ym.shape
    assert rel_fro(ya, ym) <= 5e-6, name


def _ref_cls(mod, cls):
    """The class from the REFERENCE checkout: the drop-in package exports the same import paths, so its modules are moved out of the
    way for the import and restored afterwards (tests/test_oracle_golden.py does the same)."""
    sys.dont_write_bytecode = True
    tops = ("attention_mechanisms", "vision_transformers", "mlps")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in tops}
    sys.path.insert(0, REFERENCE)
    try:
        m = importlib.import_module(mod)
        assert m.__file__.startswith(REFERENCE), "not the reference's module: %s" % m.__file__
        return getattr(m, cls)
    finally:
        sys.path.remove(REFERENCE)
        for k in list(sys.modules):
            if k.split(".")[0] in tops:
                del sys.modules[k]
        sys.modules.update(saved)


REF = {"se": ("attention_mechanisms.se_module", "SELayer", (64,), {}), "eca": ("attention_mechanisms.eca", "ECALayer", (64,), {}),
       "cbam": ("attention_mechanisms.cbam", "CBAM", (64,), {}), "da": ("attention_mechanisms.double_attention", "DoubleAttention", (64, 32, 32), {}),
       "vit_attn": ("vision_transformers.ViT", "Attention", (192, 6), {}), "vit_enc": ("vision_transformers.ViT", "TransformerEncoder", (192, 6), {}),
       "vit": ("vision_transformers.ViT", "VisionTransformer", (), dict(image_size=32, patch_size=8, depths=2, num_heads=4, embedding_dim=64, num_classes=10)),
       "xca": ("vision_transformers.xcit", "XCA", (96, 4), dict(qkv_bias=True)),
       "xca_block": ("vision_transformers.xcit", "XCABlock", (96, 4), dict(qkv_bias=True, eta=1.0)),
       "mixer": ("mlps.mlp_mixer", "MixerLayer", (64, 49), {})}


@pytest.mark.skipif(not have_reference(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_aten_sequence_equals_live_reference(name):
    """Same operators in the same order as the reference module: agreement to fp32 rounding noise of the threaded kernels."""
    _, _, shape, f_aten, _ = next(c for c in CASES if c[0] == name)
    mod, cls, args, kw = REF[name]
    ref_cls = _ref_cls(mod, cls)
    torch.manual_seed(1234)
    m = ref_cls(*args, **kw).eval()
    torch.manual_seed(4321)
    x = torch.randn(*shape)
    with torch.no_grad():
        yr = m(x, 6, 6) if name == "xca_block" else m(x)
        ya = f_aten(x, _sd(m))
    assert rel_fro(ya, yr) <= 2e-6, name
