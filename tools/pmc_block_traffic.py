#!/usr/bin/env python
"""Fabric traffic per forward of ONE bench block from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of
`bench.py --workload all --only <block>`:

    python tools/pmc_block_traffic.py <block name> <fetch.db> <write.db> <forwards>  ->  one JSON line {block, bytes, kernels}

Every kernel of the run except torch's own (RNG fill, copies) and the stream-copy yardstick belongs to the block; `forwards` is the
number of forward calls of the run (warmup + steps + 1 + steps of bench.py).  Units/corrections as in tools/pmc_traffic.py
(MI355X_MICROARCH.md section HBM): KiB counters, FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream);
Infinity-Cache hits are included -- traffic below L2, not DRAM-only.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
    return {re.sub(r"\(anonymous namespace\)::", "", n): (cnt, tot) for n, cnt, tot in rows}


def main(block, fetch_db, write_db, forwards):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    total, kernels = 0.0, {}
    for name in sorted(set(f) | set(w)):
        if "at::" in name or "rocclr" in name or "stream_copy" in name or "elementwise_kernel" in name:
            continue
        cnt, ft = f.get(name, (0, 0.0))
        _, wt = w.get(name, (0, 0.0))
        b = (2.0 * ft + wt) * 1024.0 / forwards
        kernels[re.sub(r"^void ", "", name)[:90]] = {"launches_per_forward": round(cnt / forwards, 2), "bytes_per_forward": int(b)}
        total += b
    print(json.dumps({"block": block, "bytes": int(total), "kernels": kernels}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]))
