#!/usr/bin/env python
"""Record the pre-projection Fourier position features of the REAL reference (vision_transformers/xcit.py:56-74) for a few grids:
fp64 checksums and strided samples -> tests/golden/fourier.json.  Run in the build container (needs /root/reference); the GPU box and
the CPU suite only read the json.  The features are what PositionalEncodingFourier.forward feeds to its 1x1 conv: they depend on
(H, W, hidden_dim, temperature) only."""
import json
import math
import os
import sys

import torch

REF = os.environ.get("MI355_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from vision_transformers.xcit import PositionalEncodingFourier  # noqa: E402


def reference_features(H, W, hidden, temperature):
    """The body of the reference forward up to (not including) token_projection, executed by the reference's own module code path:
    a 1x1 conv with identity weights returns the features themselves."""
    m = PositionalEncodingFourier(hidden_dim=hidden, dim=2 * hidden, temperature=temperature)
    with torch.no_grad():
        m.token_projection.weight.copy_(torch.eye(2 * hidden).reshape(2 * hidden, 2 * hidden, 1, 1))
        m.token_projection.bias.zero_()
        pos = m(1, H, W)                                   # (1, 2*hidden, H, W)
    return pos[0].permute(1, 2, 0).reshape(H * W, 2 * hidden).contiguous()


out = {}
for H, W, hidden, temp in ((14, 14, 32, 10000), (7, 9, 32, 10000), (3, 5, 8, 100), (24, 24, 16, 10000)):
    f = reference_features(H, W, hidden, temp)
    flat = f.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 65).long()
    out[f"{H}x{W}_h{hidden}_t{temp}"] = dict(H=H, W=W, hidden=hidden, temperature=temp, sum=float(flat.double().sum()),
                                             abssum=float(flat.double().abs().sum()), idx=idx.tolist(),
                                             samples=[float(v) for v in flat[idx].double()])
here = os.path.dirname(os.path.abspath(__file__))
json.dump(out, open(os.path.join(here, "fourier.json"), "w"), indent=1)
print("wrote", len(out), "grids")
