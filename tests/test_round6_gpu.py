"""Round 6 ("Next round" items 2 and 10):

  * reference behaviour on large activations WITHOUT caller edits: `module(x)` of a drop-in whose fp16 operands saturate returns the
    strict-mode result with one warning (option "range_fallback" = 1, the default; mi355_range_arm / mi355_range_wait), the fused
    block kernels (mlp_fused.hip, cswin_fused.hip) report into the range word, "range_fallback" = 0 keeps the round-3 contract
    (Mi355RangeError on the next call);
  * the SURVEY 8(b) spellings of three entry points are exported and run the same code;
  * the single-read exchange kernels (SE / CBAM) under contention: two launches concurrently on two streams, and beside a filler
    kernel that keeps the matrix pipes of every CU busy -- the result equals the oracle, never garbage plus a late MI355_ESYNC;
  * a ragged batch gathered over a gloo group from device tensors (two ranks sharing the GPU).
"""
import ctypes
import os
import subprocess
import sys
import warnings

import pytest
import torch

import oracle as O
from conftest import ROOT, assert_parity

pytestmark = pytest.mark.gpu


def _seeded(ctor, seed=1234):
    torch.manual_seed(seed)
    return ctor().eval()


def _sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


# ---- range fallback -------------------------------------------------------------------------------------------------------------------
def _cases():
    from mi355attn.modules import Attention, CSWinBlock, TransformerEncoder

    def attn():                      # stand-alone Attention: the fp32 input itself is cast to fp16 (mi355_cast16_fwd reports, code 1)
        m = _seeded(lambda: Attention(768, 12))
        return m, (4, 197, 768), 1e5, lambda x, sd: O.vit_attention_forward(x, sd, 12), ()

    def enc_input():                 # x * 1e3 through a pre-LN block: LayerNorm absorbs the scale, nothing saturates, no warning
        m = _seeded(lambda: TransformerEncoder(768, 12))
        return m, (4, 197, 768), 1e3, lambda x, sd: O.vit_encoder_forward(x, sd, 12), ()

    def enc_weights():               # fc1 weights x 1e5: the 16-bit hidden activation saturates in the GEMM epilogue (code 3)
        m = _seeded(lambda: TransformerEncoder(768, 12))
        with torch.no_grad():
            m.mlp.fc1.weight.mul_(1e5)
        return m, (4, 197, 768), 1.0, lambda x, sd: O.vit_encoder_forward(x, sd, 12), ()

    def cswin_mlp():                 # CSWin stage 1: gelu(H) inside the fused proj + MLP kernel (code 4, round 6)
        m = _seeded(lambda: CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True))
        with torch.no_grad():
            m.mlp.fc1.weight.mul_(1e5)
        return m, (2, 3136, 64), 1.0, lambda x, sd: O.cswin_block_forward(x, sd, 56, 2, 1), ()

    def cswin_v():                   # CSWin stage 1: the v third of the qkv projection x 1e5 -> v / ctx inside the stripe kernel (code 4)
        m = _seeded(lambda: CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True))
        with torch.no_grad():
            m.qkv.weight[128:].mul_(1e5)
        return m, (2, 3136, 64), 1.0, lambda x, sd: O.cswin_block_forward(x, sd, 56, 2, 1), ()

    def cswin_input():               # x * 1e3 into the LN-fronted block: no saturation, fast result inside 1e-3
        m = _seeded(lambda: CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True))
        return m, (2, 3136, 64), 1e3, lambda x, sd: O.cswin_block_forward(x, sd, 56, 2, 1), ()

    return {"attn_x1e5": (attn, True), "encoder_x1e3": (enc_input, False), "encoder_fc1_1e5": (enc_weights, True),
            "cswin_s1_fc1_1e5": (cswin_mlp, True), "cswin_s1_v_1e5": (cswin_v, True), "cswin_s1_x1e3": (cswin_input, False)}


def _oracle_has(name):
    return hasattr(O, name)


@pytest.mark.parametrize("case", ["attn_x1e5", "encoder_x1e3", "encoder_fc1_1e5", "cswin_s1_fc1_1e5", "cswin_s1_v_1e5", "cswin_s1_x1e3"])
def test_module_call_falls_back_to_strict_like_the_reference(case):
    """ViT.py:79-89,116-119 / cswin.py:176-197 return finite numbers at any scale.  Zero-edit drop-in, default options."""
    import mi355attn
    build, fires = _cases()[case]
    if case.startswith("encoder") and not _oracle_has("vit_encoder_forward"):
        pytest.skip("oracle has no encoder restatement")
    m, shape, xscale, ref_fn, fwd_args = build()
    sd = _sd(m)
    torch.manual_seed(4321)
    x = torch.randn(*shape) * xscale
    ref = ref_fn(x, sd)
    assert torch.isfinite(ref).all()
    md, xd = m.cuda(), x.cuda()
    assert mi355attn.get_option("range_fallback") == 1, "the default"
    mi355attn.range_status(wait=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            y = md(xd, *fwd_args)
        torch.cuda.synchronize()
    hits = [i for i in w if "re-running this forward in strict mode" in str(i.message)]
    if fires:
        assert len(hits) == 1, [str(i.message) for i in w]
        assert_parity(y.cpu(), ref, 2e-4, case + " [strict re-run]")
    else:
        assert not hits, [str(i.message) for i in hits]
        assert_parity(y.cpu(), ref, 1e-3, case + " [fast path, nothing saturated]")
    mi355attn.range_status(wait=True)                       # nothing left pending for the next caller
    # a second call behaves the same (the arm / wait state is per forward)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        with torch.no_grad():
            y2 = md(xd, *fwd_args)
    assert torch.equal(y, y2)


def test_range_fallback_off_keeps_the_round3_contract():
    """"range_fallback" = 0: no wait, the forward returns whatever fp16 produced and the NEXT 16-bit launch raises Mi355RangeError."""
    import mi355attn
    from mi355attn.modules import Attention
    m = _seeded(lambda: Attention(768, 12)).cuda()
    torch.manual_seed(4321)
    x = (torch.randn(4, 197, 768) * 1e5).cuda()
    mi355attn.range_status(wait=True)
    mi355attn.set_option("range_fallback", 0)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                y = m(x)
            torch.cuda.synchronize()
        assert not [i for i in w if "strict mode" in str(i.message)]
        assert not torch.isfinite(y).all()
        with pytest.raises(mi355attn.Mi355RangeError):
            with torch.no_grad():
                m(x)
    finally:
        mi355attn.set_option("range_fallback", 1)
        try:
            mi355attn.range_status(wait=True)
        except mi355attn.Mi355RangeError:
            pass


def test_range_wait_does_not_wait_for_launches_behind_the_last_producer():
    """mi355_range_wait synchronises on ONE event behind the last fp16 producer.  With the prediction "the first producer is the last"
    (mi355_range_arm(1 + 1)) the event is recorded in front of the first launch behind the cast16: with ~4 ms of non-reporting copies queued
    behind it the wait returns while they are still running.  Without a prediction (arm(1)) the event goes to the tail: the same wait
    drains the stream -- correct, just without the slack."""
    import mi355attn
    from mi355attn import functional as F
    L = mi355attn.lib()
    dev = torch.device("cuda", 0)
    x = torch.randn(1 << 20, device=dev)
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)        # 1 GiB: the copy behind the producer takes ~0.5 ms
    dst = torch.empty_like(big)
    torch.cuda.synchronize()
    busy = {}
    for arm in (2, 1):
        assert L.mi355_range_arm(arm) == 0
        try:
            F.cast16(x, 1)                                             # the one producer
            for _ in range(8):
                F.stream_copy(big, dst)                                # ~4 ms of non-reporting work behind it
            done = torch.cuda.Event()
            done.record()
            assert L.mi355_range_wait() == 0
            busy[arm] = not done.query()
            assert L.mi355_range_launches() == 1
        finally:
            L.mi355_range_arm(0)
        torch.cuda.synchronize()
    assert busy[2], "with the prediction, mi355_range_wait must not drain the stream"
    assert not busy[1], "without a prediction the event is recorded at the tail"


def test_survey_8b_aliases_run_the_same_code():
    import mi355attn
    from mi355attn import _ffi
    L = mi355attn.lib()
    torch.manual_seed(3)
    qkv = torch.randn(2, 197, 3 * 768, device="cuda")
    a, b = torch.empty(2, 197, 768, device="cuda"), torch.empty(2, 197, 768, device="cuda")
    st = _ffi.stream_ptr(qkv.device)
    assert L.mi355_sdpa_fwd(_ffi.dptr(qkv), _ffi.dptr(a), 2, 197, 12, 64, ctypes.c_float(0.125), 1, st) == 0
    assert L.mi355_sdpa_core_fwd(_ffi.dptr(qkv), _ffi.dptr(b), 2, 197, 12, 64, ctypes.c_float(0.125), 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    x, w, bias = torch.randn(64, 256, device="cuda"), torch.randn(128, 256, device="cuda"), torch.randn(128, device="cuda")
    y0, y1 = torch.empty(64, 128, device="cuda"), torch.empty(64, 128, device="cuda")
    for fn, y in ((L.mi355_linear_fwd, y0), (L.mi355_gemm_bias_act_fwd, y1)):
        assert fn(_ffi.dptr(x), _ffi.dptr(w), _ffi.dptr(bias), None, None, _ffi.dptr(y), 64, 128, 256, 256, 128, 1, 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert L.mi355_mixer_token_mlp_workspace_bytes(4, 196, 512) == L.mi355_mixer_token_workspace_bytes(4, 196, 512) > 0
    assert L.mi355_xca_workspace_bytes(2, 196, 8, 48) == 0 and L.mi355_layernorm_workspace_bytes(10, 64) == 0
    assert L.mi355_cswin_lepe_attn_workspace_bytes(2, 56, 64) == 0 and L.mi355_sdpa_core_workspace_bytes(2, 197, 12, 64) == 0


# ---- exchange kernels under contention ------------------------------------------------------------------------------------------------
def _chan_mods(C=256):
    from mi355attn.modules import CBAM, SELayer
    torch.manual_seed(1234)
    return SELayer(C).eval(), CBAM(C).eval()


@pytest.mark.parametrize("filler", [False, True], ids=["two_streams", "two_streams_plus_mfma_filler"])
def test_single_read_exchange_kernels_under_contention(filler):
    """SE and CBAM single-read launches poll for granules of peer workgroups of the SAME image.  Two such launches run concurrently on
    two streams (each sized for the whole chip), optionally beside a register-operand MFMA loop (two 4-wave workgroups on every CU)
    on a third stream.  Every output must equal the oracle; no MI355_ESYNC may be pending afterwards."""
    import mi355attn
    from mi355attn import functional as F
    se, cb = _chan_mods()
    sds = _sd(se), _sd(cb)
    torch.manual_seed(99)
    xa, xb = torch.randn(48, 256, 56, 56), torch.randn(48, 256, 56, 56)
    ref = {"se_a": O.se_forward(xa[:2], sds[0]["fc.0.weight"], sds[0]["fc.2.weight"]),
           "cb_b": O.cbam_forward(xb[:2], sds[1]["ca.fc.0.weight"], sds[1]["ca.fc.2.weight"], sds[1]["sa.conv.weight"]),
           "cb_a": O.cbam_forward(xa[-2:], sds[1]["ca.fc.0.weight"], sds[1]["ca.fc.2.weight"], sds[1]["sa.conv.weight"]),
           "se_b": O.se_forward(xb[-2:], sds[0]["fc.0.weight"], sds[0]["fc.2.weight"])}
    se, cb = se.cuda(), cb.cuda()
    se2, cb2 = _chan_mods()
    se2, cb2 = se2.cuda(), cb2.cuda()                               # second module instances: their own exchange workspaces per stream anyway
    xa, xb = xa.cuda(), xb.cuda()
    mi355attn.sync_status(wait=True)
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    sink, rep = torch.zeros(4, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        se(xa), cb(xb), se2(xb), cb2(xa)                               # code objects loaded, workspaces zeroed
    torch.cuda.synchronize()
    outs = []
    for it in range(6):
        if filler:
            with torch.cuda.stream(s3):                                # ~10 ms of MFMA work on every CU
                F.check(mi355attn.lib().mi355_mfma_yardstick(0, 1 << 17, F.dptr(sink), F.dptr(rep), F.stream_ptr(sink.device)), "yardstick")
        with torch.no_grad():
            with torch.cuda.stream(s1):
                ya = se(xa)
                ca = cb(xa)
            with torch.cuda.stream(s2):
                cbb = cb2(xb)
                yb = se2(xb)
        outs.append((ya, ca, cbb, yb))
    torch.cuda.synchronize()
    mi355attn.sync_status(wait=True)                                   # raises if any exchange ran out of its poll budget
    for ya, ca, cbb, yb in outs:
        assert_parity(ya[:2].cpu(), ref["se_a"], 1e-5, "SE stream 1")
        assert_parity(ca[-2:].cpu(), ref["cb_a"], 1e-5, "CBAM stream 1")
        assert_parity(cbb[:2].cpu(), ref["cb_b"], 1e-5, "CBAM stream 2")
        assert_parity(yb[-2:].cpu(), ref["se_b"], 1e-5, "SE stream 2")
    first = outs[0]
    for o in outs[1:]:
        assert all(torch.equal(p, q) for p, q in zip(first, o)), "results differ between contended repetitions"


# ---- ragged gather over gloo from device tensors ----------------------------------------------------------------------
_RAGGED = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join({root!r}, "pytorch-attention_amd"))
from mi355attn.dist import forward_sharded
from mi355attn.modules import SELayer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
torch.manual_seed(1234)
m = SELayer(64).eval().cuda()
torch.manual_seed(4321)
x = torch.randn(5, 64, 16, 16, device="cuda")                # 5 images over 2 ranks: 3 + 2
with torch.no_grad():
    y = forward_sharded(m, x)
    full = m(x)
assert y.is_cuda and tuple(y.shape) == tuple(full.shape) and torch.equal(y, full), (y.shape, full.shape)
dist.barrier()
dist.destroy_process_group()
print("RAGGED_OK", rank)
"""


def test_ragged_gather_over_gloo_with_device_tensors(tmp_path):
    script = tmp_path / "ragged.py"
    script.write_text(_RAGGED.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29733", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.count("RAGGED_OK") == 2, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_comm_selftest_passes_and_times_out_without_hanging(monkeypatch):
    """bench.make_comm: the C-ABI communicator self-test runs on a helper thread with a deadline; the verdict is an all-reduce over the
    torch.distributed group.  One rank, nccl backend: the real communicator passes; a construction that never returns is reported as a
    time-out and the caller falls back (on an 8-GPU node a stuck RCCL bootstrap must not hang the scaling run)."""
    import socket
    import time
    import torch.distributed as dist
    import bench
    import mi355attn.dist as mdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        comm, why = bench.make_comm(dist, dev, 0, 1, timeout_s=60.0)
        assert comm is not None and why is None, why
        got = comm.all_gather(torch.ones(2, 3, device=dev))
        assert got.shape == (2, 3)
        comm.close()

        class Stuck:
            def __init__(self, *a, **k):
                time.sleep(30)
        monkeypatch.setattr(mdist, "RcclComm", Stuck)
        t0 = time.time()
        comm, why = bench.make_comm(dist, dev, 0, 1, timeout_s=1.0)
        assert comm is None and "timed out" in why and time.time() - t0 < 10
    finally:
        dist.destroy_process_group()
