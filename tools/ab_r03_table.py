#!/usr/bin/env python
"""Collect gpurun_out/ab_*/ (tools/ab_r03.sh, one directory per lease) into a markdown table: r03 build vs HEAD, same box, per lease.

    python tools/ab_r03_table.py gpurun_out/ab_* > profiles/r05_ab_r03_vs_head.md
"""
import glob
import json
import os
import re
import sys


def op_times(path):
    base, new = [], []
    for line in open(path):
        m = re.search(r"'base': ([\d.]+), 'new': ([\d.]+)", line)
        if m:
            base.append(float(m.group(1)))
            new.append(float(m.group(2)))
    return (min(base), min(new)) if base else (None, None)


def block_ms(path):
    try:
        line = [l for l in open(path) if l.startswith("{")][-1]
    except (OSError, IndexError):
        return None
    d = json.loads(line)
    blocks = d.get("blocks") or d.get("config", {}).get("blocks") or []
    return blocks[0]["ms"] if blocks else None


def main(dirs):
    print("| lease | copy GB/s | MFMA 16x16x32 / 32x32x16 TF | sclk MHz (issue) | " +
          " | ".join("%s r03 / HEAD us" % o for o in ("proj", "fc2", "fc1", "qkv")) + " | " +
          " | ".join("%s r03 / HEAD ms" % b for b in ("XCABlock", "MixerLayer", "ViT-Base")) + " |")
    print("|" + "---|" * 11)
    for d in dirs:
        y = {}
        try:
            y = json.loads([l for l in open(os.path.join(d, "yardstick.json")) if l.startswith("{")][-1])
        except (OSError, IndexError, ValueError):
            pass
        cells = [os.path.basename(d), str(y.get("stream_copy_GBps", "?")),
                 "%s / %s" % (y.get("mfma_16x16x32_TFLOPs", "?"), y.get("mfma_32x32x16_TFLOPs", "?")), str(y.get("sclk_MHz_issue", "?"))]
        for op in ("proj", "fc2", "fc1", "qkv"):
            b, n = op_times(os.path.join(d, "op_%s.txt" % op)) if os.path.exists(os.path.join(d, "op_%s.txt" % op)) else (None, None)
            cells.append("%s / %s" % (b, n))
        for blk in ("XCABlock", "MixerLayer", "VisionTransformer"):
            r = [block_ms(p) for p in sorted(glob.glob(os.path.join(d, "r03_%s_*.json" % blk)))]
            h = [block_ms(p) for p in sorted(glob.glob(os.path.join(d, "head_%s_*.json" % blk)))]
            r, h = [v for v in r if v], [v for v in h if v]
            cells.append("%s / %s" % (min(r) if r else "?", min(h) if h else "?"))
        print("| " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
