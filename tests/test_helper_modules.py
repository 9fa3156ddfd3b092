"""CPU checks of host-side pieces added in round 3: the Fourier position features in the engine's own closed form against samples
recorded from the real reference (tests/golden/fourier.json, made by tests/golden/make_fourier.py) and, when the reference checkout
is present, against the live reference bit for bit; dropout rates are accepted the way the reference accepts them."""
import json
import os

import pytest
import torch

from conftest import REFERENCE, ROOT


def _features(H, W, hidden, temperature):
    from mi355attn.modules.xcit import PositionalEncodingFourier
    m = PositionalEncodingFourier(hidden_dim=hidden, dim=2 * hidden, temperature=temperature)
    return m.features(H, W)


def test_fourier_features_match_recorded_reference_samples():
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "fourier.json")))
    assert len(rec) >= 4
    for name, r in rec.items():
        f = _features(r["H"], r["W"], r["hidden"], r["temperature"]).reshape(-1)
        assert f.numel() == r["H"] * r["W"] * 2 * r["hidden"]
        got = f[torch.tensor(r["idx"])].double()
        assert torch.equal(got, torch.tensor(r["samples"], dtype=torch.float64)), f"{name}: samples differ from the reference"
        assert float(f.double().sum()) == r["sum"] and float(f.double().abs().sum()) == r["abssum"], name


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
def test_fourier_features_bit_identical_to_live_reference():
    import importlib
    import sys
    sys.dont_write_bytecode = True
    # the reference's packages carry the same names as the drop-in import-path shims: import, then restore what was loaded before
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("attention_mechanisms", "vision_transformers", "mlps")}
    sys.path.insert(0, REFERENCE)
    try:
        Ref = importlib.import_module("vision_transformers.xcit").PositionalEncodingFourier
    finally:
        sys.path.remove(REFERENCE)
        for k in list(sys.modules):
            if k.split(".")[0] in ("attention_mechanisms", "vision_transformers", "mlps"):
                del sys.modules[k]
        sys.modules.update(saved)
    for H, W, hidden, temp in ((14, 14, 32, 10000), (5, 11, 8, 50), (1, 1, 4, 10000), (31, 2, 6, 1000)):
        m = Ref(hidden_dim=hidden, dim=2 * hidden, temperature=temp)
        with torch.no_grad():
            m.token_projection.weight.copy_(torch.eye(2 * hidden).reshape(2 * hidden, 2 * hidden, 1, 1))
            m.token_projection.bias.zero_()
            ref = m(2, H, W)[1].permute(1, 2, 0).reshape(H * W, 2 * hidden)
        assert torch.equal(_features(H, W, hidden, temp), ref), (H, W, hidden, temp)


def test_dropout_rates_are_accepted_as_eval_identity():
    """A model built with its training configuration (attn_drop / proj_drop > 0) must construct; in eval mode nn.Dropout is the
    identity in the reference too.  Only the stochastic training-mode forward is refused (checked on the GPU suite's side as well)."""
    from mi355attn.modules import mhsa
    m = mhsa.Attention(64, num_heads=2, attn_drop=0.1, proj_drop=0.2)
    assert m.attn_drop.p == 0.1 and m.proj_drop.p == 0.2
    with pytest.raises(ValueError):
        mhsa.Attention(64, num_heads=2, attn_drop=1.5)
    m.train()
    with pytest.raises(RuntimeError, match="eval"):
        m(torch.randn(1, 4, 64))
