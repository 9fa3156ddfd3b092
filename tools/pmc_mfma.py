#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (rocpd SQLite .db), for the ViT MHSA block (BASELINE configs[2]).

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d out -o c3 -- python bench.py --no-cpu --workload c3 ...
    python tools/pmc_mfma.py out/c3_results.db

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs): the fraction of all matrix-pipe cycles of the chip that
were busy while the kernel ran (the gfx94x MfmaUtil formula; ROCm 7.2 has no gfx950 section in its derived-counter files).  Next to
it the tool prints what the kernel's arithmetic needs at the dense rate (16x16x32 f16/bf16 MFMA = 16 passes x 4 cycles... i.e.
2*16*16*32 / 1024 FLOP per SIMD cycle = 16 cycles per instruction) when --flops-per-launch is given.
"""
import re
import sqlite3
import sys

SIMDS = 256 * 4


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
    per = {}
    for name, cn, cnt, tot in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        per.setdefault(name, {})[cn] = (cnt, tot)
    print(f"{'launches':>8} {'MFMA busy cyc/launch':>22} {'GUI active cyc/launch':>22} {'MFMA util':>10}  kernel")
    for name, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]):
        if "at::" in name or "rocclr" in name:
            continue
        n, busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))
        _, gui = d.get("GRBM_GUI_ACTIVE", (0, 0.0))
        if not n:
            continue
        util = busy / (gui * SIMDS) if gui else float("nan")
        print(f"{n:8d} {busy / n:22.0f} {gui / n:22.0f} {util:10.3f}  {name[:90]}")
    others = sorted({cn for d in per.values() for cn in d} - {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
    for cn in others:
        print(f"\n{cn} per launch:")
        for name, d in per.items():
            if cn in d and "at::" not in name and "rocclr" not in name:
                print(f"  {d[cn][1] / d[cn][0]:16.0f}  {name[:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
