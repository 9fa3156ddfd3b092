"""How far inside the 1e-3 bar is the default fp16-operand mode, and how does the margin move with the seed and the weight scale?

VERDICT round 5 (weak #1): every golden / full-size / smoke case used seeds 1234 / 4321 and the module-default init, so the margin
was shown at one point.  Here C3 (`Attention(768, 12)`, ViT.py:79-89) and C5 (`VisionTransformer(num_heads=12)`, ViT.py:180-192,
init `:147-158`) run at B = 16 for 5 weight seeds x 5 input seeds (paired) x weight scale {1, 2, 3} (every parameter with >= 2
dimensions multiplied by the scale: the Linear / conv kernels, cls token and position rows -- the init's std), fast (fp16 operands)
and strict (split-bf16) against the fp32 oracle.  The table (rel-Frobenius and max-abs ratio per cell, both modes) is written to
$MI355_MARGIN_OUT (default gpurun_out/r06_parity_margin.md) and committed as profiles/r06_parity_margin.md.

Asserted: at scale 1 every seed passes both criteria at 1e-3 (fast) and 5e-5 (strict).  At scales 2 and 3 the strict mode must
stay fp32-class (<= 1e-4: the oracle's own fp32 rounding grows with the logit scale) and the fast result must be finite and within
3e-3 -- a cell above 1e-3 there is DOCUMENTED in the table with the strict figure beside it (the fallback a caller has), not
hidden: 3x the init std is outside what the reference's init produces (`trunc_normal_(std=.02)`).
"""
import os

import pytest
import torch

import oracle as O
from conftest import ROOT, max_abs_ratio, rel_fro

pytestmark = pytest.mark.gpu

SEEDS = [(1234, 4321), (1, 2), (7, 11), (2024, 930), (31337, 271828)]
SCALES = [1.0, 2.0, 3.0]
B = 16
_rows = []


def _scaled(ctor, seed, scale):
    torch.manual_seed(seed)
    m = ctor().eval()
    if scale != 1.0:
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() >= 2:
                    p.mul_(scale)
    return m


def _cell(name, ctor, shape, ref_fn, wseed, xseed, scale):
    import mi355attn
    m = _scaled(ctor, wseed, scale)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(xseed)
    x = torch.randn(*shape)
    ref = ref_fn(x, sd)
    md = m.cuda()
    xd = x.cuda()
    with torch.no_grad():
        y = md(xd).cpu()
    old = mi355attn.default_precision()
    mi355attn.set_default_precision(0)
    try:
        with torch.no_grad():
            ys = md(xd).cpu()
    finally:
        mi355attn.set_default_precision(old)
    rec = dict(block=name, wseed=wseed, xseed=xseed, scale=scale, fast_fro=rel_fro(y, ref), fast_max=max_abs_ratio(y, ref),
               strict_fro=rel_fro(ys, ref), strict_max=max_abs_ratio(ys, ref), finite=bool(torch.isfinite(y).all()))
    _rows.append(rec)
    print("[margin] %-8s wseed %-6d xseed %-6d scale %.0fx  fast rel_fro %.2e max_abs %.2e | strict rel_fro %.2e max_abs %.2e"
          % (name, wseed, xseed, scale, rec["fast_fro"], rec["fast_max"], rec["strict_fro"], rec["strict_max"]))
    return rec


def _judge(rec):
    assert rec["finite"], rec
    if rec["scale"] == 1.0:
        assert rec["fast_fro"] <= 1e-3 and rec["fast_max"] <= 1e-3, rec
        assert rec["strict_fro"] <= 5e-5 and rec["strict_max"] <= 5e-5, rec
    else:
        assert rec["strict_fro"] <= 1e-4 and rec["strict_max"] <= 1e-4, rec
        assert rec["fast_fro"] <= 3e-3 and rec["fast_max"] <= 3e-3, rec


@pytest.mark.parametrize("scale", SCALES)
@pytest.mark.parametrize("seeds", SEEDS, ids=["w%d" % s[0] for s in SEEDS])
def test_c3_margin(seeds, scale):
    from mi355attn.modules import Attention
    _judge(_cell("C3", lambda: Attention(768, 12), (B, 197, 768), lambda x, sd: O.vit_attention_forward(x, sd, 12), seeds[0], seeds[1], scale))


@pytest.mark.parametrize("scale", SCALES)
@pytest.mark.parametrize("seeds", SEEDS, ids=["w%d" % s[0] for s in SEEDS])
def test_c5_margin(seeds, scale):
    from mi355attn.modules import VisionTransformer
    _judge(_cell("C5", lambda: VisionTransformer(num_heads=12), (B, 3, 224, 224), lambda x, sd: O.vit_forward(x, sd, 12, 12),
                 seeds[0], seeds[1], scale))


def test_zz_write_margin_table():
    """Runs last in this file: the table of every cell measured above (also when some of them failed)."""
    if not _rows:
        pytest.skip("no cell ran")
    path = os.environ.get("MI355_MARGIN_OUT", os.path.join(ROOT, "gpurun_out", "r06_parity_margin.md"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write("# Parity margin of the default fp16-operand mode: 5 seeds x weight scale {1, 2, 3}, B = %d (tests/test_parity_margin_gpu.py)\n\n" % B)
            f.write("Bar: rel-Frobenius <= 1e-3 AND max|y-r| <= 1e-3 max|r| (SURVEY 8d); strict bar 5e-5.  Oracle: fp32 torch-CPU restatement.\n\n")
            f.write("| block | weight seed | input seed | weight scale | fast rel-fro | fast max-abs | strict rel-fro | strict max-abs | inside 1e-3 |\n")
            f.write("|---|---|---|---|---|---|---|---|---|\n")
            for r in _rows:
                ok = r["fast_fro"] <= 1e-3 and r["fast_max"] <= 1e-3
                f.write("| %s | %d | %d | %.0fx | %.2e | %.2e | %.2e | %.2e | %s |\n" % (
                    r["block"], r["wseed"], r["xseed"], r["scale"], r["fast_fro"], r["fast_max"], r["strict_fro"], r["strict_max"],
                    "yes" if ok else "NO (strict is the fallback)"))
            for blk in ("C3", "C5"):
                for s in SCALES:
                    sel = [r for r in _rows if r["block"] == blk and r["scale"] == s]
                    if sel:
                        f.write("\n%s scale %.0fx: worst fast rel-fro %.2e, worst fast max-abs %.2e (margin %.1fx / %.1fx inside 1e-3); worst strict %.2e / %.2e\n" % (
                            blk, s, max(r["fast_fro"] for r in sel), max(r["fast_max"] for r in sel),
                            1e-3 / max(r["fast_fro"] for r in sel), 1e-3 / max(r["fast_max"] for r in sel),
                            max(r["strict_fro"] for r in sel), max(r["strict_max"] for r in sel)))
    except OSError as e:
        pytest.skip("table not written: %s" % e)
