"""CPU tests (-m "not gpu"): pin the oracle.

1. Every oracle function reproduces the golden outputs that tests/golden/make_golden.py recorded from the
   REAL reference (fp64 checksums, the SURVEY.md 8(c) probe values, 257 strided samples, full tensors for
   the small cases).  Weights and inputs are regenerated from the seed protocol through the drop-in
   modules' constructors (no reference checkout needed).
2. The drop-in modules reproduce the reference's state_dict layout and init stream (per-parameter shapes and
   fp64 checksums match the golden record).
3. When the reference checkout is present (build container only) the oracle is additionally compared with
   the live reference on shapes outside the golden table (ragged HW, odd channel counts, batch 1).
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from cases import BY_ID, CASES, build_case, flat_out, sample_index
from conftest import REFERENCE, ROOT, have_reference, rel_fro


def _build(c):
    cls = getattr(importlib.import_module(c["mod"]), c["cls"])
    return build_case(c, cls)


CASE_IDS = [c["id"] for c in CASES]


@pytest.mark.parametrize("cid", CASE_IDS)
def test_dropin_params_match_reference_init(cid, golden):
    c, g = BY_ID[cid], golden[cid]
    m, x = _build(c)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["params"].keys()), "state_dict keys/order differ from the reference"
    for k, rec in g["params"].items():
        assert list(sd[k].shape) == rec["shape"], k
        assert float(sd[k].double().sum()) == pytest.approx(rec["sum"], rel=1e-12, abs=1e-12), k
        assert float(sd[k].double().abs().sum()) == pytest.approx(rec["abs_sum"], rel=1e-12, abs=1e-12), k
    assert float(x.double().sum()) == pytest.approx(g["x_sum"], rel=1e-12, abs=1e-12)


@pytest.mark.parametrize("cid", CASE_IDS)
def test_oracle_matches_golden(cid, golden):
    c, g = BY_ID[cid], golden[cid]
    m, x = _build(c)
    y = flat_out(c["oracle"](x, m.state_dict(), torch.float32))
    assert list(y.shape) == g["y_shape"]
    yf = y.reshape(-1)
    n = yf.numel()
    samples = torch.tensor(g["samples"], dtype=torch.float64)
    got = yf[sample_index(n)].double()
    scale = samples.abs().max()
    assert float((got - samples).abs().max()) <= 2e-5 * float(scale), "strided samples differ from the reference"
    probe = torch.tensor([float(yf[0]), float(yf[n // 3]), float(yf[-1])], dtype=torch.float64)
    assert float((probe - torch.tensor(g["probe"], dtype=torch.float64)).abs().max()) <= 2e-5 * float(scale)
    assert float(yf.double().abs().sum()) == pytest.approx(g["abs_sum"], rel=2e-6)
    assert abs(float(yf.double().sum()) - g["sum"]) <= 2e-6 * g["abs_sum"]


@pytest.mark.parametrize("cid", [c["id"] for c in CASES if c.get("small")])
def test_oracle_matches_full_small_tensor(cid):
    c = BY_ID[cid]
    z = np.load(os.path.join(ROOT, "tests", "golden", "small", cid + ".npz"))
    m, x = _build(c)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), z["p:" + k]), f"parameter {k} differs from the reference init"
    ref = torch.from_numpy(z["y"])
    y32 = flat_out(c["oracle"](x, m.state_dict(), torch.float32))
    y64 = flat_out(c["oracle"](x, m.state_dict(), torch.float64))
    assert rel_fro(y32, ref) <= 2e-6
    assert rel_fro(ref, y64) <= 2e-6          # the reference itself sits this close to the fp64 truth


def test_eca_kernel_size_rule():
    import oracle as O
    assert [O.eca_kernel_size(c) for c in (16, 64, 128, 256, 512, 1024)] == [3, 3, 5, 5, 5, 5]
    from mi355attn.modules import ECALayer
    for c in (16, 64, 256, 1000):
        assert ECALayer(c).conv.weight.shape[-1] == O.eca_kernel_size(c)


def test_explicit_stencils_match_aten_convs():
    """The oracle calls ATen conv1d/conv2d (what the reference's modules run); tap-by-tap loops agree with them."""
    import oracle as O
    import torch.nn.functional as TF
    torch.manual_seed(0)
    pooled, wk = torch.randn(3, 37, dtype=torch.float64), torch.randn(5, dtype=torch.float64)
    assert torch.allclose(O.eca_gate_explicit(pooled, wk), TF.conv1d(pooled[:, None], wk.reshape(1, 1, 5), padding=2)[:, 0],
                          atol=1e-12)
    smap, w7 = torch.randn(2, 2, 9, 11, dtype=torch.float64), torch.randn(1, 2, 7, 7, dtype=torch.float64)
    assert torch.allclose(O.spatial_conv_explicit(smap, w7), TF.conv2d(smap, w7, padding=3)[:, 0], atol=1e-12)


def test_window_index_is_a_permutation_and_bit_exact():
    import oracle as O
    for reso, hs, ws in ((56, 56, 1), (56, 1, 56), (28, 28, 2), (28, 2, 28), (14, 14, 7), (14, 7, 14), (7, 7, 7)):
        tab = O.window_token_index(reso, hs, ws)
        assert sorted(tab.reshape(-1).tolist()) == list(range(reso * reso))


# ---- live cross-checks against the real reference (build container only) ----------------------------------
needs_ref = pytest.mark.skipif(not have_reference(), reason="reference checkout not present")


def _ref_cls(mod, cls):
    sys.dont_write_bytecode = True
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in ("attention_mechanisms", "vision_transformers", "mlps")}
    sys.path.insert(0, REFERENCE)
    try:
        return getattr(importlib.import_module(mod), cls)
    finally:
        sys.path.remove(REFERENCE)
        for k in list(sys.modules):
            if k.split(".")[0] in ("attention_mechanisms", "vision_transformers", "mlps"):
                del sys.modules[k]
        sys.modules.update(saved)


@needs_ref
@pytest.mark.parametrize("shape,C,red", [((1, 48, 7, 9), 48, 16), ((3, 32, 5, 5), 32, 4), ((2, 80, 13, 1), 80, 16)])
def test_oracle_vs_live_reference_ragged_channel_attention(shape, C, red):
    import oracle as O
    torch.manual_seed(7)
    x = torch.randn(*shape)
    se = _ref_cls("attention_mechanisms.se_module", "SELayer")(C, red).eval()
    cb = _ref_cls("attention_mechanisms.cbam", "CBAM")(C, red, 3).eval()
    ec = _ref_cls("attention_mechanisms.eca", "ECALayer")(C).eval()
    with torch.no_grad():
        assert rel_fro(O.se_forward(x, se.fc[0].weight, se.fc[2].weight), se(x)) <= 1e-6
        assert rel_fro(O.cbam_forward(x, cb.ca.fc[0].weight, cb.ca.fc[2].weight, cb.sa.conv.weight), cb(x)) <= 1e-6
        assert rel_fro(O.eca_forward(x, ec.conv.weight), ec(x)) <= 1e-6


@needs_ref
def test_oracle_vs_live_reference_lepe_all_modes():
    import oracle as O
    Lepe = _ref_cls("vision_transformers.cswin", "LePEAttention")
    torch.manual_seed(3)
    for reso, idx, split, dim, heads in ((8, 0, 2, 32, 2), (8, 1, 4, 32, 1), (6, -1, 6, 64, 4)):
        m = Lepe(dim, reso, idx, split_size=split, num_heads=heads).eval()
        qkv = torch.randn(3, 2, reso * reso, dim)
        with torch.no_grad():
            ref = m(qkv)
        got = O.lepe_attention_forward(qkv, m.get_v.weight, m.get_v.bias, reso, idx, split, heads)
        assert rel_fro(got, ref) <= 1e-6


@pytest.mark.parametrize("dim,reso,heads,split,last", [(64, 56, 2, 1, False), (128, 28, 4, 2, False), (256, 14, 8, 7, False), (512, 7, 16, 7, True),
                                                     (32, 8, 2, 2, False), (64, 6, 2, 3, False)])
def test_cswin_aten_sequence_restatement_equals_the_index_table_oracle(dim, reso, heads, split, last):
    """bench.py times the CSWin CPU leg with the ATen-operator-sequence restatement (oracle/cswin.py cswin_block_forward_aten); it
    must be the same function as the auditable index-table oracle the parity tests use."""
    import oracle as O
    torch.manual_seed(11)
    C = dim
    p = {"norm1.weight": torch.rand(C) + 0.5, "norm1.bias": torch.randn(C) * 0.1, "norm2.weight": torch.rand(C) + 0.5, "norm2.bias": torch.randn(C) * 0.1,
         "qkv.weight": torch.randn(3 * C, C) / C ** 0.5, "qkv.bias": torch.randn(3 * C) * 0.1, "proj.weight": torch.randn(C, C) / C ** 0.5,
         "proj.bias": torch.randn(C) * 0.1, "mlp.fc1.weight": torch.randn(4 * C, C) / C ** 0.5, "mlp.fc1.bias": torch.randn(4 * C) * 0.1,
         "mlp.fc2.weight": torch.randn(C, 4 * C) / (4 * C) ** 0.5, "mlp.fc2.bias": torch.randn(C) * 0.1}
    cb = C if (last or reso == split) else C // 2
    for i in range(1 if (last or reso == split) else 2):
        p[f"attns.{i}.get_v.weight"] = torch.randn(cb, 1, 3, 3) * 0.3
        p[f"attns.{i}.get_v.bias"] = torch.randn(cb) * 0.1
    x = torch.randn(3, reso * reso, C)
    a = O.cswin_block_forward_aten(x, p, reso, heads, split, last)
    b = O.cswin_block_forward(x, p, reso, heads, split, last)
    assert rel_fro(a, b) <= 1e-6
