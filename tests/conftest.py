"""pytest configuration: path setup, the `gpu` marker, shared parity helpers."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-attention_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = os.environ.get("MI355_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libmi355attn.so exists (builds it with hipcc when missing) and return its path."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi355_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def rel_fro(y, r):
    y, r = y.double().cpu(), r.double().cpu()
    return float((y - r).norm() / r.norm().clamp_min(1e-300))


def max_abs_ratio(y, r):
    y, r = y.double().cpu(), r.double().cpu()
    return float((y - r).abs().max() / r.abs().max().clamp_min(1e-300))


def assert_parity(y, r, tol=1e-3, what=""):
    """PASS iff ||y-r||_F/||r||_F <= tol AND max|y-r| <= tol * max|r|."""
    assert tuple(y.shape) == tuple(r.shape), f"{what}: shape {tuple(y.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(y).all(), f"{what}: non-finite output"
    rf, ma = rel_fro(y, r), max_abs_ratio(y, r)
    assert rf <= tol and ma <= tol, f"{what}: rel_fro={rf:.3e} max_abs_ratio={ma:.3e} (tol {tol:g})"
    return rf, ma


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "attention_mechanisms"))


class no_range_fallback:
    """The round-3 boundary contract for the body: "range_fallback" = 0 on the current device (module(x) does not wait for its fp16
    producers and does not re-run; a saturated operand is reported by the NEXT call / mi355_range_status).  Default since round 6: 1."""

    def __enter__(self):
        import mi355attn
        self.old = mi355attn.get_option("range_fallback")
        mi355attn.set_option("range_fallback", 0)
        return self

    def __exit__(self, *exc):
        import mi355attn
        mi355attn.set_option("range_fallback", self.old)
        return False
