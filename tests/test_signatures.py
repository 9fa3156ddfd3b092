"""Drop-in boundary: constructor and forward signatures of EVERY mirrored class equal the reference's, positional
order included.  tests/golden/signatures.json is generated from the real reference by tests/golden/make_signatures.py; the
only difference a drop-in may have is one trailing `precision=None` keyword (the MFMA operand-format selector)."""
import importlib
import inspect
import json
import os
import re

import pytest

from conftest import ROOT, have_reference

with open(os.path.join(ROOT, "tests", "golden", "signatures.json")) as f:
    RECORD = json.load(f)

ITEMS = [(m, k) for m, v in sorted(RECORD.items()) for k in sorted(v)]


def _norm(sig):
    """Signature text without a trailing precision keyword, spacing and float spelling normalised (0.0 == 0.)."""
    s = re.sub(r" at 0x[0-9a-f]+", "", sig)
    s = re.sub(r",\s*precision=None\)", ")", s)
    s = re.sub(r"\(self,\s*precision=None\)", "(self)", s)
    s = s.replace("<class 'torch.nn.modules.activation.GELU'>", "GELU").replace("<class 'torch.nn.modules.normalization.LayerNorm'>", "LayerNorm")
    return re.sub(r"\s+", " ", s)


def _params(sig_text):
    """[(name, default text or None)] parsed from a normalised signature string (good enough for the plain signatures here)."""
    inner = sig_text[sig_text.index("(") + 1:sig_text.rindex(")")]
    out, depth, cur = [], 0, ""
    for ch in inner:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    res = []
    for p in out:
        name, _, default = p.partition("=")
        res.append((name.strip(), default.strip() or None))
    return res


def _same_default(a, b):
    if a == b:
        return True
    try:
        return float(a) == float(b)
    except (TypeError, ValueError):
        return False


@pytest.mark.parametrize("mod,name", ITEMS, ids=["%s.%s" % it for it in ITEMS])
def test_signature_matches_reference(mod, name):
    want = RECORD[mod][name]
    obj = getattr(importlib.import_module(mod), name)
    assert os.path.abspath(inspect.getfile(obj)).startswith(os.path.join(ROOT, "pytorch-attention_amd")), "not the drop-in"
    if want["kind"] == "class":
        assert inspect.isclass(obj)
        got = str(inspect.signature(obj.__init__))
    else:
        got = str(inspect.signature(obj))
    gp, wp = _params(_norm(got)), _params(_norm(want["init"]))
    assert [n for n, _ in gp] == [n for n, _ in wp], f"{mod}.{name}: parameter order {got} vs reference {want['init']}"
    for (n, dg), (_, dw) in zip(gp, wp):
        assert (dg is None) == (dw is None) and (dg is None or _same_default(dg, dw)), \
            f"{mod}.{name}: default of `{n}` is {dg}, reference has {dw}"
    if "forward" in want and want["kind"] == "class":
        fg, fw = _params(_norm(str(inspect.signature(obj.forward)))), _params(_norm(want["forward"]))
        # the drop-in may accept extra OPTIONAL trailing arguments (fused residual); the reference's own must match in order
        assert [n for n, _ in fg][:len(fw)] == [n for n, _ in fw], f"{mod}.{name}.forward: {fg} vs reference {fw}"
        assert all(d is not None for _, d in fg[len(fw):]), f"{mod}.{name}.forward: extra required arguments {fg[len(fw):]}"


@pytest.mark.skipif(not have_reference(), reason="reference checkout not present")
def test_record_is_current():
    """The committed record equals what make_signatures.py produces from the live reference (guards against a stale file)."""
    import subprocess
    import sys
    import tempfile
    import shutil
    src = os.path.join(ROOT, "tests", "golden", "signatures.json")
    keep = tempfile.mktemp()
    shutil.copy(src, keep)
    try:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_signatures.py")], check=True, env=env,
                       capture_output=True, timeout=600)
        with open(src) as f:
            assert json.load(f) == RECORD
    finally:
        shutil.move(keep, src)


@pytest.mark.parametrize("mod", ["vision_transformers.pvt", "vision_transformers.cmt"])
def test_sr_ratio_is_third_positional(mod):
    """pvt.py:98 / cmt.py:119 call `Attention(dim, num_heads, sr_ratio, ...)` positionally."""
    A = importlib.import_module(mod).Attention
    m = A(64, 1, 8)
    assert m.sr_ratio == 8 and m.num_heads == 1 and m.q.bias is None and hasattr(m, "sr")
    m = A(64, 2, 1, True)
    assert m.sr_ratio == 1 and m.q.bias is not None and not hasattr(m, "sr")
