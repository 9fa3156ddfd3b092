#!/usr/bin/env python
"""profiles/r01_bench_workloads.md from the bench lines tools/gpu_round.sh leaves in gpurun_out/ (bench_c2.json, bench_others.jsonl)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(out):
    lines = [json.loads(open(os.path.join(ROOT, "gpurun_out", "bench_c2.json")).read().strip().splitlines()[-1])]
    with open(os.path.join(ROOT, "gpurun_out", "bench_others.jsonl")) as f:
        lines += [json.loads(l) for l in f if l.strip().startswith("{")]
    rows = ["# Per-block results on one MI355X, B=256 per GPU (tools/gpu_round.sh; default fp16-operand mode; HIP-event time per block forward)",
            "", "| workload | block | ms / call | images/s | achieved | frac of roofline |", "|---|---|---|---|---|---|"]
    for d in lines:
        for b in d["config"]["blocks"]:
            rows.append("| %s | %s | %s | %s | %s %s | %s |" % (d["config"]["workload"][:44], b["block"], b["ms"], b["images_per_s"],
                                                             b["achieved"], b["unit"], b["frac"]))
    rows += ["", "Step-level lines (value = images/s through the whole step):", "", "```"]
    for d in lines:
        d = dict(d)
        d.pop("config", None)
        rows.append(json.dumps(d))
    rows.append("```")
    open(out, "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_bench_workloads.md"))
