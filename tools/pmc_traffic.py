#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of `bench.py --workload c2` into per-block
fabric traffic per forward, and write profiles/pmc_traffic.json (read back by bench.py as roofline.traffic).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch_c2/c2_results.db gpurun_out/pmc_write_c2/c2_results.db

Units/corrections (MI355X_MICROARCH.md section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced stream (confirmed here on the float4 copy kernel: 822 MB copied -> FETCH 401 421 KiB,
WRITE 802 816 KiB), so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-Cache hits are counted (these are
L2-miss / fabric requests), so this is traffic below L2, not DRAM-only traffic.
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BLOCK_OF = [  # (regex on the kernel name, block, share of that kernel's launches belonging to the block)
    (r"se_single_kernel", "SELayer(256)", 1.0),
    (r"eca_halo_kernel", "ECALayer(256)", 1.0),
    (r"cbam_single_kernel", "CBAM(256)", 1.0),
    (r"gate_scale_kernel<0", "SELayer(256)", 1.0),
    (r"gate_scale_kernel<1", "ECALayer(256)", 1.0),
    (r"pool_rows_kernel<false", "SELayer(256)", 0.5),
    (r"pool_rows_kernel<false", "ECALayer(256)", 0.5),
    (r"pool_rows_kernel<true", "CBAM(256)", 1.0),
    (r"cbam_(spatial|apply)", "CBAM(256)", 1.0),
]


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name=? group by name",
                     (counter,)).fetchall()
    return {re.sub(r"\(anonymous namespace\)::", "", n): (cnt, tot) for n, cnt, tot in rows}


def main(fetch_db, write_db, forwards_per_block):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    blocks = {}
    table = []
    for name in sorted(set(f) | set(w)):
        if "at::" in name or "rocclr" in name or "stream_copy" in name:
            continue
        cnt, ft = f.get(name, (0, 0.0))
        _, wt = w.get(name, (0, 0.0))
        total_bytes = (2.0 * ft + wt) * 1024.0
        table.append((name[:70], cnt, 2 * ft * 1024 / max(cnt, 1), wt * 1024 / max(cnt, 1)))
        for rx, blk, share in BLOCK_OF:
            if re.search(rx, name):
                blocks[blk] = blocks.get(blk, 0.0) + share * total_bytes / forwards_per_block
    print(f"{'launches':>8} {'read B/launch':>16} {'write B/launch':>16}  kernel")
    for n, cnt, rb, wb in table:
        print(f"{cnt:8d} {rb:16.0f} {wb:16.0f}  {n}")
    print()
    for k, v in blocks.items():
        print(f"{k}: {v/1e9:.3f} GB per forward  ({v / (2*256*256*56*56*4):.2f} x algorithmic)")
    out = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(out)) if os.path.exists(out) else {}
    data["c2"] = {k: round(v) for k, v in blocks.items()}
    data["_note"] = "bytes below L2 per block forward = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the block's kernels; " \
                    "Infinity-Cache hits included; separate rocprofv3 --pmc passes of bench.py --workload c2"
    json.dump(data, open(out, "w"), indent=1)


if __name__ == "__main__":
    # bench.py --no-cpu --steps 3 --warmup 1: each block runs warmup + steps + 1 + steps forwards
    fw = int(sys.argv[3]) if len(sys.argv) > 3 else (1 + 3 + 1 + 3)
    main(sys.argv[1], sys.argv[2], fw)
