#!/usr/bin/env python
"""bench.py -- forward throughput of the MI355X hot path, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|mixer|da] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] ("c2"): SELayer + CBAM + ECALayer, x = (256,256,56,56) fp32 per GPU.
One "step" = one forward of each block of the workload over the per-GPU batch (inputs resident in HBM,
H2D excluded).  `value` = images/s through the whole step, aggregated over all ranks (weak scaling: per-GPU
batch fixed, batch-sharded, no data-path collective for block workloads; c5 all-gathers the logits over RCCL).
`roofline` describes the dominant (slowest) block of the step: achieved = algorithmic bytes (or FLOPs) of
that block / its average duration measured with HIP events on the launch stream.
`cpu_baseline` = the oracle (torch-CPU restatement of the reference forward) timed on this host's cores on a
bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA


def _seeded(ctor, seed=1234):
    torch.manual_seed(seed)
    return ctor().eval()


# ------------------------------------------------------------------------------------------------------------
# workloads: each returns dict(name, blocks, gather, dtype); a block = dict(name, module, x, fwd_args,
#            bound "hbm"|"mfma", work = algorithmic bytes or FLOPs per call, cpu(callable on a host sample))
# ------------------------------------------------------------------------------------------------------------
def workload_c2(B, dev):
    from mi355attn.modules import CBAM, ECALayer, SELayer
    import oracle as O
    C, H, W = 256, 56, 56
    torch.manual_seed(4321)
    x = torch.randn(B, C, H, W, device=dev)
    nbytes = 2 * B * C * H * W * 4                      # read x once + write y once
    se, cb, ec = _seeded(lambda: SELayer(C)), _seeded(lambda: CBAM(C)), _seeded(lambda: ECALayer(C))

    def cpu_se(xs, m=se):
        return O.se_forward(xs, m.fc[0].weight.cpu(), m.fc[2].weight.cpu())

    def cpu_cb(xs, m=cb):
        return O.cbam_forward(xs, m.ca.fc[0].weight.cpu(), m.ca.fc[2].weight.cpu(), m.sa.conv.weight.cpu())

    def cpu_ec(xs, m=ec):
        return O.eca_forward(xs, m.conv.weight.cpu())

    blocks = [
        dict(name="SELayer(256)", module=se.to(dev), x=x, bound="hbm", work=nbytes, cpu=cpu_se),
        dict(name="CBAM(256)", module=cb.to(dev), x=x, bound="hbm", work=nbytes, cpu=cpu_cb),
        dict(name="ECALayer(256)", module=ec.to(dev), x=x, bound="hbm", work=nbytes, cpu=cpu_ec),
    ]
    return dict(name="SELayer+CBAM+ECALayer fwd, x=(%d,256,56,56) fp32 per GPU (BASELINE configs[1])" % B,
                blocks=blocks, gather=None, dtype="f32")


WORKLOADS = {"c2": workload_c2}


def _extra_workloads():
    try:
        import bench_workloads  # noqa: F401  (registers c3/c4/c5/... when present)
        WORKLOADS.update(bench_workloads.WORKLOADS)
    except ImportError:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=64, help="images in the CPU-baseline sample")
    ap.add_argument("--chunk-images", type=int, default=None, help="override the Infinity-Cache chunk size")
    ap.add_argument("--nt", type=int, default=None, help="channel-attention final pass: bit0 NT loads, bit1 NT stores")
    ap.add_argument("--reverse", type=int, default=None, help="channel-attention final pass walks the batch backwards")
    ap.add_argument("--precision", type=int, default=None, help="MFMA operand precision 0 strict / 1 fp16 / 2 bf16")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="mi355_set_option override (tuning experiments)")
    ap.add_argument("--only", default=None, help="keep only the blocks of the workload whose name contains this text (profiling aid)")
    args = ap.parse_args()
    _extra_workloads()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mi355attn
    from mi355attn import StreamTimer
    if args.chunk_images is not None:
        mi355attn.set_option("chunk_images", args.chunk_images)
    if args.nt is not None:
        mi355attn.set_option("nt", args.nt)
    if args.reverse is not None:
        mi355attn.set_option("reverse", args.reverse)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        mi355attn.set_option(key, int(val))
    if args.precision is not None:
        mi355attn.set_default_precision(args.precision)

    wl = WORKLOADS[args.workload](args.batch, dev)
    wname, blocks, gather = wl["name"], wl["blocks"], wl.get("gather")
    if args.only:
        blocks = [b for b in blocks if args.only in b["name"]]
        wname += " [only: %s]" % args.only
        if not blocks:
            raise SystemExit("--only matched no block")

    def step():
        outs = []
        with torch.no_grad():
            for b in blocks:
                outs.append(b["module"](b["x"], *b.get("fwd_args", ())))
        if gather and dist is not None:
            outs = [gather(o, dist) for o in outs]
        return outs

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-block durations with HIP events on the launch stream (un-timed extra passes)
    per_block = []
    for b in blocks:
        with torch.no_grad():
            b["module"](b["x"], *b.get("fwd_args", ()))
        torch.cuda.synchronize()
        tm = StreamTimer(dev)
        tm.start()
        with torch.no_grad():
            for _ in range(args.steps):
                b["module"](b["x"], *b.get("fwd_args", ()))
        ms = tm.stop_ms() / args.steps
        if b["bound"] == "hbm":
            ach, peak, unit = b["work"] / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = b["work"] / (ms * 1e-3) / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
        per_block.append(dict(block=b["name"], ms=round(ms, 4), images_per_s=round(args.batch / (ms * 1e-3), 1),
                              bound=b["bound"], achieved=round(ach, 2), peak=peak, unit=unit, frac=round(ach / peak, 4)))

    # achievable-bandwidth yardstick: float4 streaming copy of the same footprint
    copy_gbs = None
    if rank == 0:
        from mi355attn import functional as F
        src = blocks[0]["x"].reshape(-1)
        if src.numel() * 4 % 16 == 0 and src.numel() * 4 >= (1 << 26):
            dst = torch.empty_like(src)
            F.stream_copy(src, dst)
            torch.cuda.synchronize()
            tm = StreamTimer(dev)
            tm.start()
            for _ in range(10):
                F.stream_copy(src, dst)
            cms = tm.stop_ms() / 10
            copy_gbs = round(2 * src.numel() * 4 / (cms * 1e-3) / 1e9, 1)
            del dst

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed
    dom = max(per_block, key=lambda r: r["ms"])
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get(args.workload, {}).get(dom["block"])
        except Exception:
            traffic = None
    out = {
        "metric": "forward images/sec (+ ms/block), B=%d per GPU, 224x224-derived shapes" % args.batch,
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if wl.get("dtype", "f32") == "f32" else {0: "bf16x3", 1: "f16", 2: "bf16"}[mi355attn.default_precision()],
        "data": "synthetic (torch.randn seed 4321; module-default init seed 1234)",
        "config": {"workload": wname, "batch_per_gpu": args.batch, "parallelism": "batch-shard x%d" % world,
                   "chunk_images": mi355attn.get_option("chunk_images"), "nt": mi355attn.get_option("nt"),
                   "reverse": mi355attn.get_option("reverse"),
                   "precision": {0: "strict(bf16x3)", 1: "fp16-mfma", 2: "bf16-mfma"}[mi355attn.default_precision()],
                   "blocks": per_block, "stream_copy_GBps": copy_gbs},
        "roofline": {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                     "frac": dom["frac"], "traffic": traffic, "kernel": dom["block"], "ms": dom["ms"]},
    }

    if world == 1 and not args.no_cpu:
        ncores = os.cpu_count() or 1
        ns = min(args.cpu_sample, args.batch)
        # pick the torch thread count that is fastest on this host (all SMT threads is often NOT it on big dual-socket boxes)
        probe = blocks[0]
        xs0 = probe["x"][:ns].cpu()
        best_t, best_n = None, None
        for nthr in sorted({min(ncores, n) for n in (8, 16, 32, 64, 128, ncores)}):
            torch.set_num_threads(nthr)
            probe["cpu"](xs0)
            t1 = time.perf_counter()
            probe["cpu"](xs0)
            dt = time.perf_counter() - t1
            if best_t is None or dt < best_t:
                best_t, best_n = dt, nthr
        torch.set_num_threads(best_n)
        t_cpu = 0.0
        reps = 3
        for b in blocks:
            xs = b["x"][:ns].cpu() if b["x"].shape[0] >= ns else b["x"].cpu()
            b["cpu"](xs)                                           # warm-up
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                b["cpu"](xs)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            t_cpu += ts[len(ts) // 2]
        out["cpu_baseline"] = {"value": round(ns / t_cpu, 1), "unit": "images/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": "oracle (torch-CPU restatement of the reference forward) on the first %d images "
                                         "of the same batch, median of %d after 1 warm-up, all blocks of the step" % (ns, reps)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
