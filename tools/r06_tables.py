#!/usr/bin/env python
"""Round-6 tables from the evidence set of tools/gpu_round6_final.sh (gpurun_out/r6f):

    python tools/r06_tables.py next   > profiles/r06_next_rows.md      # next rows on HEAD at B = 256
    python tools/r06_tables.py blocks                                  # the per-block table of DESIGN.md 6 / 7 (markdown, stdout)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r6f")


def line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def pmc(path):
    out = {}
    if os.path.exists(path):
        for l in open(path):
            l = l.strip()
            if l:
                r = json.loads(l)
                out[r["block"]] = r
    return out


def next_rows():
    traffic = pmc(os.path.join(O, "pmc_next.jsonl"))
    print("# Next rows (SURVEY 8 f1-f3) on HEAD at B = 256, one MI355X -- round 6 (VERDICT round 5, item 8)\n")
    print("`bench.py --workload <w> --no-cpu --no-calib` (20 steps, 5 warm-up; per-block HIP events on the launch stream), PMC traffic = (2 x FETCH_SIZE +")
    print("WRITE_SIZE) x 1 KiB per block forward from separate `rocprofv3 --pmc` passes of `--only <block>` (`tools/gpu_round6_final.sh`).  frac = of 2.5 PFLOP/s")
    print("(MFMA-graded) or 8 TB/s (HBM-graded); algorithmic work as in `bench_workloads.py` (x in + y out for the gates; projections + QK^T + PV for f1).\n")
    for wl, title in (("cswin", "f3: CSWin-T/224 full forward (cswin.py:360-363)"), ("xcit", "f3: XCiT-nano-12/16 full forward (xcit.py:416-420)"),
                      ("mixer_full", "f3: MLP-Mixer(512, depth 12) full forward (mlp_mixer.py:65-79)"),
                      ("f1", "f1: plain-MHSA copies at their native stage shapes"), ("zoo", "f2: SimAM, SRM, Gaussian GCT, LCT, GCT at the C2 shape"),
                      ("zoo2", "f2: GC, CoordAtt, Triplet, BAM, SK, CAM at the C2 shape")):
        p = os.path.join(O, "next_%s.json" % wl)
        if not os.path.exists(p):
            continue
        try:
            d = line(p)
        except Exception as e:                                # noqa: BLE001
            print("## %s\n\nno line: %s\n" % (title, e))
            continue
        print("## %s\n" % title)
        print("step: %.4f ms = %.0f images/s\n" % (d["ms_per_step"], d["value"]))
        print("| block | ms | achieved | frac | strict ms | PMC bytes / forward | images/s |")
        print("|---|---|---|---|---|---|---|")
        for b in d["blocks"]:
            t = traffic.get("%s: %s" % (wl, b["block"]), {}).get("bytes")
            print("| %s | %.4f | %s %s | %.4f | %s | %s | %.0f |" % (
                b["block"], b["ms"], b["achieved"], b["unit"], b["frac"], ("%.3f" % b["strict_ms"]) if b.get("strict_ms") else "-",
                ("%.3f GB" % (t / 1e9)) if t else "-", 256.0 / (b["ms"] * 1e-3)))
        print()


def blocks():
    d = line(os.path.join(O, "bench_all.json"))
    print("| block | ms | graded roof: achieved | frac | HBM side: algorithmic / counter bytes, of 8 TB/s | counter / algorithmic | strict ms (x fast) |")
    print("|---|---|---|---|---|---|---|")
    hb = dict(kv.split("=") for kv in d["roofline"]["hbm_fracs"].split(",") if kv)
    tx = dict(kv.split("=") for kv in d["roofline"]["traffic_x"].split(",") if kv)
    for b in d["blocks"]:
        k = b["key"]
        print("| %s | %.4f | %s %s | %.4f | %s | %s | %s |" % (
            b["block"], b["ms"], b["achieved"], b["unit"], b["frac"], hb.get(k, "(graded on HBM)"), tx.get(k, "-"),
            ("%.3f (%.2f)" % (b["strict_ms"], b["strict_ms"] / b["ms"])) if b.get("strict_ms") else "-"))
    c = d["config"]
    print("\nstep %.4f ms = %.1f images/s; windows %s; box: copy %s GB/s, MFMA %s / %s TFLOP/s (16x16x32 / 32x32x16), %s MHz and %s W under load" % (
        d["ms_per_step"], d["value"], c["ms_windows"], c["stream_copy_GBps"], c["mfma_16x16x32_TFLOPs"], c["mfma_32x32x16_TFLOPs"],
        c["sclk_MHz_load"], c["power_W_load"]))
    cb = d["cpu_baseline"]
    print("cpu_baseline %.2f images/s on %s cores (%s); ViT-Base %s images/s; %s" % (cb["value"], cb["cores"], cb["host"], cb.get("img_s_ViTBase"), cb.get("GFLOPs")))
    r = d["roofline"]
    print("roofline: %s" % {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "block", "kernel", "kernel_avg_us", "kernel_frac")})


if __name__ == "__main__":
    {"next": next_rows, "blocks": blocks}[sys.argv[1]]()
