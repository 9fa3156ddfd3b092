#!/usr/bin/env python
"""Record the constructor / forward signatures of every reference class the drop-in package mirrors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_signatures.py [--ref /root/reference]

Runs in the build container only (imports the REAL reference).  For every shim module under
pytorch-attention_amd/{attention_mechanisms,vision_transformers,mlps,cnns}/ it imports the reference module of the same
path and, for every public class or factory function the shim exports that the reference module defines too, writes
`str(inspect.signature(...))` of the constructor (or function) and of `forward` into tests/golden/signatures.json.
tests/test_signatures.py diffs the drop-in classes against that record (the only allowed difference is a trailing
`precision=None` keyword).
"""
import argparse
import importlib
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, "pytorch-attention_amd")
DIRS = ("attention_mechanisms", "vision_transformers", "mlps", "cnns")


def shim_modules():
    out = []
    for d in DIRS:
        for f in sorted(os.listdir(os.path.join(PKG, d))):
            if f.endswith(".py") and f != "__init__.py":
                out.append(d + "." + f[:-3])
    return out


def public_names(mod):
    import torch
    names = []
    for k, v in vars(mod).items():
        if k.startswith("_"):
            continue
        if inspect.isclass(v) and issubclass(v, torch.nn.Module) and v.__module__.split(".")[0] != "torch":
            names.append(k)
        elif inspect.isfunction(v):
            names.append(k)
    return sorted(names)


def _sig(fn):
    import re
    return re.sub(r" at 0x[0-9a-f]+", "", str(inspect.signature(fn)))          # function-valued defaults print their address


def describe(obj):
    if inspect.isclass(obj):
        d = {"kind": "class", "init": _sig(obj.__init__)}
        if "forward" in vars(obj) or any("forward" in vars(b) for b in obj.__mro__[1:] if b.__module__.split(".")[0] != "torch"):
            d["forward"] = _sig(obj.forward)
        return d
    return {"kind": "function", "init": _sig(obj)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import _stub_timm
    _stub_timm()
    # 1. names the shims export (imported from the drop-in package)
    sys.path.insert(0, PKG)
    exported = {}
    for name in shim_modules():
        exported[name] = public_names(importlib.import_module(name))
    for name in list(sys.modules):
        if name.split(".")[0] in DIRS:
            del sys.modules[name]
    sys.path.remove(PKG)
    # 2. the same names in the real reference
    sys.path.insert(0, args.ref)
    record = {}
    import contextlib
    import io
    for name, names in exported.items():
        with contextlib.redirect_stdout(io.StringIO()):          # setr.py runs its smoke block at import time
            ref = importlib.import_module(name)
        assert os.path.abspath(ref.__file__).startswith(os.path.abspath(args.ref)), ref.__file__
        for k in names:
            if hasattr(ref, k):
                record.setdefault(name, {})[k] = describe(getattr(ref, k))
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(record, f, indent=1, sort_keys=True)
    print("wrote", sum(len(v) for v in record.values()), "signatures from", len(record), "modules")


if __name__ == "__main__":
    main()
