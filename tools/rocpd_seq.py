#!/usr/bin/env python
"""Dispatch-ordered kernel list of the LAST forward in a rocprofv3 rocpd result (kernel-trace) of `bench.py --only <block>`:
every launch of one block forward with its duration, grid and workgroup size -- per-kernel attribution of a block's time (a
--stats table merges launches of one kernel on different shapes).

    python tools/rocpd_seq.py <results.db> <launches_per_forward | 0 = detect the period> [title]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    return name if len(name) < 100 else name[:97] + "..."


def main(path, per, title):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    sel = "name, start, end" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 0")
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    rows = [r for r in rows if not any(t in r[0] for t in ("at::", "rocclr", "stream_copy", "elementwise_kernel"))]
    names = [r[0] for r in rows]
    if not rows:
        print(f"# {title}: no kernel launches in {path}")
        return
    if per <= 0:                                    # smallest period of the tail of the launch sequence
        n = len(names)
        per = n
        for p in range(1, n // 2 + 1):
            if names[n - p:] == names[n - 2 * p:n - p] and (n < 3 * p or names[n - 3 * p:n - 2 * p] == names[n - p:]):
                per = p
                break
    reps = 0                                        # average each position over every complete period at the tail
    n = len(rows)
    while (reps + 1) * per <= n and [r[0] for r in rows[n - (reps + 1) * per:n - reps * per]] == names[n - per:]:
        reps += 1
    print(f"# {title}: {per} launches per forward, averaged over the last {reps} forwards of the run")
    print(f"{'#':>3} {'avg_us':>9} {'min_us':>9} {'grid':>9} {'wg':>5}  kernel")
    tot = 0.0
    for i in range(per):
        ds = [(rows[n - (k + 1) * per + i][2] - rows[n - (k + 1) * per + i][1]) / 1e3 for k in range(reps)]
        r = rows[n - per + i]
        tot += sum(ds) / len(ds)
        print(f"{i:3d} {sum(ds)/len(ds):9.2f} {min(ds):9.2f} {r[3]:9d} {r[4]:5d}  {short(r[0])}")
    span = [(rows[n - k * per - 1][2] - rows[n - (k + 1) * per][1]) / 1e3 for k in range(reps)]
    print(f"    {tot:9.2f} us summed kernel time; first-start to last-end of a forward: avg {sum(span)/len(span):.2f} us, min {min(span):.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else sys.argv[1])
