#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (rocpd SQLite .db), for the ViT MHSA block (BASELINE configs[2]).

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d out -o c3 -- python bench.py --no-cpu --workload c3 ...
    python tools/pmc_mfma.py out/c3_results.db

rocpd stores one row per counter INSTANCE (32 per dispatch for the SQ counters, 8 for GRBM), so values are summed per dispatch first.
  MFMA util   = sum SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x sum SQ_BUSY_CU_CYCLES): matrix-pipe cycles busy per CU-busy cycle
                (SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per 16x16x32 f16/bf16 instruction, i.e. the dense-rate time of the kernel's MFMAs;
                 checked here: the qkv GEMM needs 2*50432*2304*768 / 16384 = 10.9 M instructions = 174 M cycles, the counter reads 174 M)
  shader clock = sum SQ_BUSY_CU_CYCLES / 256 CUs / kernel duration: the clock the CUs actually ran at while the kernel was resident
GRBM_GUI_ACTIVE is not in shader cycles on this part (about 0.5 GHz) and is only printed.
"""
import re
import sqlite3
import sys

CUS = 256


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, dispatch_id, counter_name, sum(counter_value), min(duration) from pmc_events "
                     "group by name, dispatch_id, counter_name").fetchall()
    per = {}
    for name, disp, cn, tot, dur in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        if "at::" in name or "rocclr" in name:
            continue
        d = per.setdefault(name, {})
        e = d.setdefault(disp, {"dur": dur})
        e[cn] = tot
    print(f"{'launches':>8} {'avg us':>9} {'MFMA busy Mcyc':>15} {'CU busy Mcyc':>13} {'MFMA util':>10} {'clock GHz':>10}  kernel")
    for name, disps in sorted(per.items(), key=lambda kv: -sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for e in kv[1].values())):
        n = len(disps)
        busy = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for e in disps.values()) / n
        cu = sum(e.get("SQ_BUSY_CU_CYCLES", 0.0) for e in disps.values()) / n
        dur = sum(e["dur"] for e in disps.values()) / n
        util = busy / (4.0 * cu) if cu else float("nan")
        clk = cu / CUS / dur if dur else float("nan")
        print(f"{n:8d} {dur / 1e3:9.1f} {busy / 1e6:15.2f} {cu / 1e6:13.2f} {util:10.3f} {clk:10.2f}  {name[:80]}")


if __name__ == "__main__":
    main(sys.argv[1])
