#!/usr/bin/env python
"""Generate the golden fixtures by running the REAL reference (import from /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

Only runs in the build container (the reference checkout does not exist on the GPU box).  For every case
in tests/golden/cases.py it builds the reference module under the seed protocol (oracle/params.py), runs
its CPU forward (eval, no_grad, fp32) and records into tests/golden/golden.json:
  * fp64 sum(y), sum(|y|), the three probe values of SURVEY.md 8(c) (flat[0], flat[n//3], flat[-1]),
  * 257 strided samples of y (cases.sample_index),
  * per-parameter fp64 checksums (sum, sum|.|) + shapes, so the drop-in modules can prove they reproduce
    the reference's parameter layout AND init stream without a checkpoint file.
``small`` cases additionally get the full y and every parameter in tests/golden/small/<id>.npz (x is
regenerated from the input seed and checked through ``x_sum``).
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from cases import CASES, build_case, flat_out, make_arg, sample_index  # noqa: E402


def _stub_timm():
    """vision_transformers/dilateformer.py:19-20 imports timm helpers (DropPath, to_2tuple, trunc_normal_, _cfg) that are not
    installed here.  GlobalAttention (the class the fixtures cover) uses none of them, so importing the file only needs the
    names to exist."""
    import types
    try:
        import timm  # noqa: F401
        return
    except ImportError:
        pass
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    vit = types.ModuleType("timm.models.vision_transformer")

    def _unavailable(*a, **k):
        raise RuntimeError("timm is not installed: stub used only to import dilateformer.py")

    layers.DropPath = layers.to_2tuple = layers.trunc_normal_ = _unavailable
    vit._cfg = _unavailable
    timm.models, models.layers, models.vision_transformer = models, layers, vit
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers, "timm.models.vision_transformer": vit})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    _stub_timm()
    torch.set_num_threads(os.cpu_count())
    out = {"torch": torch.__version__, "protocol": "weights seed 1234, input seed 4321, eval, fp32 CPU",
           "cases": {}}
    os.makedirs(os.path.join(HERE, "small"), exist_ok=True)
    for c in CASES:
        mod = importlib.import_module(c["mod"])
        cls = getattr(mod, c["cls"])
        m, x = build_case(c, cls)
        with torch.no_grad():
            y = flat_out(m(x, *[make_arg(a) for a in c.get("fwd_args", ())]))
        yf = y.reshape(-1)
        n = yf.numel()
        rec = {
            "x_shape": list(x.shape), "y_shape": list(y.shape),
            "sum": float(yf.double().sum()), "abs_sum": float(yf.double().abs().sum()),
            "probe": [float(yf[0]), float(yf[n // 3]), float(yf[-1])],
            "samples": [float(v) for v in yf[sample_index(n)]],
            "x_sum": float(x.double().sum()),
            "params": {k: {"shape": list(v.shape), "sum": float(v.double().sum()),
                           "abs_sum": float(v.double().abs().sum())}
                       for k, v in m.state_dict().items()},
        }
        out["cases"][c["id"]] = rec
        if c.get("small"):
            np.savez_compressed(os.path.join(HERE, "small", c["id"] + ".npz"), y=y.numpy(),
                                **{"p:" + k: v.numpy() for k, v in m.state_dict().items()})
        print(f"{c['id']:12s} sum={rec['sum']:.6f} abs={rec['abs_sum']:.6f} probe={rec['probe']}")
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
