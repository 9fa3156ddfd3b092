#!/usr/bin/env python
"""Assemble per-block PMC traffic records (tools/pmc_block_traffic.py, one JSON line per block) into the table bench.py reads:

    python tools/pmc_collect.py <blocks.jsonl> <out.json>

The table is stamped with bench.csrc_fingerprint() -- the hash of the kernel sources it was measured against; bench.py relays a
block's `traffic` only when the stamp matches the tree it runs from.  NOTE the dominant-kernel probe of bench.py adds GEMM launches to
a `--only VisionTransformer` run: their traffic is attributed separately (kernels are listed per record) and excluded here by
forward-count normalisation only for kernels that belong to the block proper -- see `excluded`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(src, dst):
    table, per_kernel = {}, {}
    for line in open(src):
        line = line.strip()
        if not line:
            continue
        r = json.loads(line)
        table[r["block"]] = r["bytes"]
        per_kernel[r["block"]] = r["kernels"]
    out = {"_csrc_sha16": bench.csrc_fingerprint(),
           "_note": "bytes below L2 per block forward = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the block's kernels; Infinity-Cache hits "
                    "included; separate rocprofv3 --pmc passes of `bench.py --only <block>` (tools/gpu_round3.sh)",
           "all": table, "kernels": per_kernel}
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "blocks:", len(table), "csrc", out["_csrc_sha16"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
