#!/usr/bin/env python
"""Per-kernel averages of every counter of a rocprofv3 --pmc pass (rocpd SQLite .db): counter instances summed per dispatch, then
averaged over the dispatches of a kernel.    python tools/pmc_dump.py out/x_results.db [name-substring]"""
import sqlite3
import sys


def main(path, sub=""):
    c = sqlite3.connect(path)
    rows = c.execute("select name, dispatch_id, counter_name, sum(counter_value), min(duration) from pmc_events "
                     "group by name, dispatch_id, counter_name").fetchall()
    per = {}
    for name, did, cn, val, dur in rows:
        if sub and sub not in name:
            continue
        k = per.setdefault(name, {})
        k.setdefault(cn, []).append(val)
        k.setdefault("_dur_us", {})[did] = dur / 1000.0
    for name, k in sorted(per.items(), key=lambda kv: -sum(kv[1]["_dur_us"].values())):
        d = k.pop("_dur_us")
        print("%s  launches=%d avg_us=%.1f" % (name[:110], len(d), sum(d.values()) / len(d)))
        for cn in sorted(k):
            print("    %-32s %14.0f" % (cn, sum(k[cn]) / len(k[cn])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
