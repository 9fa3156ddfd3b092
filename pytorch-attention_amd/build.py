#!/usr/bin/env python
"""Build libmi355attn.so (gfx950 only) in-tree with hipcc.

    python pytorch-attention_amd/build.py [--force] [--jobs N] [--keep-temps]

Compiles every csrc/*.hip to an object (in parallel, skipped when newer than its sources) and links
mi355attn/lib/libmi355attn.so.  hipcc cross-compiles without a GPU; the .so travels to the GPU box with
the tree (it is git-ignored, not gpurun-ignored).
"""
import argparse
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "mi355attn", "lib")
LIB = os.path.join(LIBDIR, "libmi355attn.so")
ARCH = "gfx950"
# -Werror=inline-asm / -Werror=pass-failed: an asm statement the compiler objects to (e.g. a reserved register on a clobber list) or a
# launch-bounds occupancy hint it cannot meet stops the build instead of scrolling by.
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=off", "-Werror=inline-asm", "-Werror=pass-failed"]
# Per-file additions.  -fno-honor-nans on the single-read CBAM kernel: its running maxima compile to v_max_f32 preceded by a
# canonicalising v_max x, x, x per operand (IEEE maxnum must quiet signalling NaNs) and the DPP permutations cannot be folded into
# them -- 190 of the kernel's 1880 VALU instructions per band, in a kernel whose bands spend 40 % of their SIMD time on VALU work.
# A NaN in x still reaches y through the average-pooling path.
EXTRA_FLAGS = {"cbam_single.hip": ["-fno-honor-nans"]}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and /opt/rocm/bin)")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, jobs=None, verbose=True, keep_temps=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mi355attn.h")]
    cc = hipcc()
    todo = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            todo.append((s, o))

    def one(job):
        s, o = job
        cmd = [cc] + CXXFLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if keep_temps:
            cmd += ["-save-temps=obj"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        return s, r

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            for s, r in ex.map(one, todo):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {s}")
                if verbose:
                    print(f"[mi355attn] compiled {os.path.basename(s)}")
    if force or todo or not _newer(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[mi355attn] linked {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--keep-temps", action="store_true")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, keep_temps=a.keep_temps)
