"""bench.py launch contract, checked on CPU (gloo): `--gpus N` must really run N ranks, whether bench.py is started bare
(it re-executes itself under torch.distributed.run) or by an external torchrun (the driver's form)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_bare_gpus2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--launch-check"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["steps"] == 3


def test_external_torchrun_form():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-check"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_no_gpu_is_a_loud_error_not_a_cpu_run():
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)


def test_host_cpu_info():
    sys.path.insert(0, ROOT)
    import bench
    cores, threads, model = bench.host_cpu_info()
    assert 1 <= cores <= threads == (os.cpu_count() or 1)
