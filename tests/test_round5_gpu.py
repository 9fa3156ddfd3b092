"""Round-5 GPU tests (VERDICT round 4 "Next round" + the advisor's findings):

  * LayerScale with the published deep-XCiT initialisation (eta = 1e-5): the branch y - x of XCABlock against fp64 (the host no longer folds
    a gamma whose products would land in the fp16 subnormals);
  * the fused Mixer token-mixing kernel reports a saturating 16-bit intermediate into the range word, like the launches it replaced;
  * its "mixer_early" epilogue variant is bit-identical to the default;
  * the in-process kernel tally: a refused launch leaves no phantom tag, a report longer than the caller's buffer is handed out whole;
  * the MFMA yardstick of the bench line;
  * `bench.py --gpus 2 --dist-backend gloo`: the whole N > 1 path (self-launch, barrier-bracketed windows, ranks_seen, host-staged
    gather) end to end with the real kernels, two ranks sharing the one visible GPU.
"""
import ctypes
import json
import os
import subprocess
import sys

import pytest
import torch

import oracle as O
from conftest import ROOT, assert_parity, rel_fro

pytestmark = pytest.mark.gpu


def _drain_range():
    import mi355attn
    try:
        mi355attn.range_status(wait=True)
    except mi355attn.Mi355RangeError:
        pass


@pytest.mark.parametrize("eta", [1e-5, 1e-3, 1.0])
def test_xcablock_small_layerscale_branch_accuracy(eta):
    """with gamma folded into fp16 weights the proj / fc2 branch lost 9e-2 relative accuracy at eta = 1e-5 and
    no test saw it (the branch is 1e-5 of the residual).  The branch y - x is compared with an fp64 evaluation of the reference math."""
    from mi355attn import functional as F
    from mi355attn.modules import XCABlock
    torch.manual_seed(1234)
    m = XCABlock(384, 8, qkv_bias=True, eta=eta).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(4, 196, 384)
    ref = O.xca_block_forward(x, sd, 8, 14, 14, dtype=torch.float64)
    m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda(), 14, 14).cpu()
    branch, want = (y.double() - x.double()), (ref.double() - x.double())
    # fp32 cancellation noise of y - x: |x| * 2^-24 against a branch of size ~eta
    floor = float(x.abs().max()) * 2.0 ** -23 / max(float(want.abs().mean()), 1e-30)
    err = rel_fro(branch, want)
    assert err <= 2e-3 + 4 * floor, f"eta={eta}: branch error {err:.3e} (cancellation floor {floor:.1e})"
    folded = F.weight16_scaled(m.attn.proj.weight, m.attn.proj.bias, m.gamma1, 1) is not None
    assert folded == (eta >= 0.1), "fold decision"


def test_fused_mixer_kernel_reports_a_saturating_intermediate():
    """LN(x) * w + b above 65504 inside mixer_token_kernel must raise Mi355RangeError at the next check."""
    import mi355attn
    from mi355attn.modules import MixerLayer
    _drain_range()
    torch.manual_seed(5)
    m = MixerLayer(256, 196, precision=1).eval().cuda()
    x = torch.randn(2, 196, 256, device="cuda")
    from conftest import no_range_fallback
    with torch.no_grad(), no_range_fallback():                     # the reporting contract itself (round 6: module(x) would absorb the report)
        tags = [t for t, *_ in mi355attn.kernel_trace(lambda: m(x))]
        assert any("mixer_token_kernel" in t for t in tags), tags
        mi355attn.range_status(wait=True)                          # ordinary data: nothing to report
        m.norm1.weight.mul_(1.0e5)                                  # LN output ~1e5: finite in fp32, inf in fp16
        raised = False
        try:
            m(x)                                                   # a later 16-bit launch of the same forward may already see the report
        except mi355attn.Mi355RangeError:
            raised = True
    if not raised:
        with pytest.raises(mi355attn.Mi355RangeError, match="fused block kernel"):
            mi355attn.range_status(wait=True)
    _drain_range()
    m2 = MixerLayer(256, 196, precision=2).eval().cuda()           # bf16 operands have the fp32 range: nothing to report
    m2.load_state_dict(m.state_dict())
    with torch.no_grad():
        m2(x)
    mi355attn.range_status(wait=True)


@pytest.mark.parametrize("prec", [1, 2])
def test_mixer_early_residual_variant_is_bit_identical(prec):
    import mi355attn
    from mi355attn.modules import MixerLayer
    torch.manual_seed(11)
    m = MixerLayer(512, 196, precision=prec).eval().cuda()
    x = torch.randn(7, 196, 512, device="cuda")
    old = mi355attn.get_option("mixer_early")
    try:
        with torch.no_grad():
            mi355attn.set_option("mixer_early", 0)
            y0 = m(x)
            mi355attn.set_option("mixer_early", 1)
            seen = []
            def run():
                seen.append(m(x))
            tags = [t for t, *_ in mi355attn.kernel_trace(run)]
    finally:
        mi355attn.set_option("mixer_early", old)
    assert any("mixer_token_kernel<early>" in t for t in tags), tags
    assert torch.equal(y0, seen[0])


def test_trace_has_no_phantom_tags_and_long_reports_survive():
    """A launch the two-accumulator kernel refuses (K too short for its fp32 epilogue) must not leave a tag; a report longer than the
    caller's buffer is kept by the library and handed out whole on the second call."""
    import mi355attn
    from mi355attn import _ffi
    from mi355attn import functional as F
    x16 = torch.randn(512, 384, device="cuda").half()
    w16 = torch.randn(256, 384, device="cuda").half()
    res = torch.randn(512, 256, device="cuda")

    def run():
        F.linear16(x16, w16, None, resid=res, precision=1)            # fp32 out + residual, K = 384 < 640: not gemm16_pa's shape
    rows = mi355attn.kernel_trace(run)
    assert len(rows) == 1 and rows[0][1] == 1, rows                   # exactly one kernel ran, exactly one tag with one launch
    assert all(mn > 0.5 for _, _, _, mn, _ in rows), rows             # no near-zero phantom interval
    # long report: many distinct tags through a 64-byte buffer
    L = _ffi.lib()
    assert L.mi355_trace_begin() == 0
    for k in (64, 128, 192, 256, 320):
        F.linear16(torch.randn(256, k, device="cuda").half(), torch.randn(64, k, device="cuda").half(), None, out16=True, precision=1)
    torch.cuda.synchronize()
    small = ctypes.create_string_buffer(64)
    need = L.mi355_trace_end(small, 64)
    assert need > 64
    big = ctypes.create_string_buffer(need + 1)
    assert L.mi355_trace_end(big, need + 1) == need
    lines = big.value.decode().splitlines()
    assert len(lines) == 5 and all(len(l.split("\t", 4)) == 5 for l in lines), lines
    assert big.value.decode().startswith(small.value.decode())
    assert L.mi355_trace_end(big, need + 1) == 0                      # handed out whole: dropped


def test_mfma_yardstick_reads_a_sane_rate_and_clock():
    from mi355attn import functional as F
    dev = torch.device("cuda", 0)
    y0 = F.mfma_yardstick(dev, 0, target_ms=10.0)
    y1 = F.mfma_yardstick(dev, 1, target_ms=10.0)
    for y in (y0, y1):
        assert 300.0 < y["TFLOPs"] < 2600.0, y                        # between a badly throttled part and the 2.5 PF nameplate
    assert 500.0 < y1["sclk_MHz_issue"] < 2500.0, y1
    if y1["sclk_MHz_counter"]:
        assert 50.0 < y1["sclk_MHz_counter"] < 3000.0, y1


@pytest.mark.parametrize("workload,first_key", [("c5", "ViTBase"), ("c2", "SE")])
def test_bench_two_ranks_share_one_gpu_on_gloo(workload, first_key):
    """c5: the end-of-forward gather path; c2 (round 6): the single-read SE / CBAM exchange kernels of two processes share the GPU -- each
    launch polls for peer workgroups of its own grid while the other rank's grid occupies CUs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", workload, "--batch", "16",
           "--steps", "2", "--warmup", "1", "--no-cpu", "--no-strict", "--no-calib"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # two ranks on ONE GPU: a launch-path run, reported as such -- n_gpus counts distinct devices
    assert line["n_gpus"] == 1 and line["scaling"].startswith("none") and line["config"]["distinct_gpus"] == 1
    assert line["config"]["ranks_seen"] == 2 and len(line["ms_per_step_by_rank"]) == 2
    assert line["config"]["dist_backend"] == "gloo" and "gloo" in line["config"]["gather"]
    assert len(line["ms_windows"]) == 3 and line["value"] > 0
    assert line["blocks"][0]["key"] == first_key


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,stride", [(256, 1000, 768, 197 * 768), (5, 10, 64, 64), (40, 33, 100, 104), (130, 257, 36, 36), (300, 700, 256, 256)])
def test_small_output_gemm_is_bit_identical_to_the_engine(M, N, K, stride, prec):
    """gemm_small.hip (one-wave 16 x 32 tiles for outputs under an eighth of a round of 128 x 128 tiles: the ViT head, VERDICT round 4
    "What's missing" 6) accumulates a row's K steps in the engine's order: same bits as gemm_kernel with the option off, also through a
    strided X (token 0 of every image in place), a ragged edge in M and N, and a K that is not a multiple of 32."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M * 7 + N)
    base = torch.randn(M, stride, device="cuda")
    x = base[:, :K]                                                   # row stride `stride`, consumed in place
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).contiguous()
    b = torch.randn(N, device="cuda")
    old = mi355attn.get_option("gemm_small")
    try:
        outs, tags = {}, {}
        for v in (1, 0):
            mi355attn.set_option("gemm_small", v)
            def run():
                outs[v] = (F.linear(x, w, b, precision=prec), F.linear(x, w, None, precision=prec))
            tags[v] = [t for t, *_ in mi355attn.kernel_trace(run)]
    finally:
        mi355attn.set_option("gemm_small", old)
    assert all("gemm_small_kernel" in t for t in tags[1]) and len(tags[1]) >= 1, tags[1]
    assert not any("gemm_small_kernel" in t for t in tags[0]), tags[0]
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    assert_parity(outs[1][0].cpu(), ref.float(), 5e-5 if prec == 0 else (1e-3 if prec == 1 else 8e-3), "small GEMM")


def test_vit_head_runs_on_the_small_tile_kernel():
    import mi355attn
    from mi355attn.modules import VisionTransformer
    torch.manual_seed(1)
    m = VisionTransformer(image_size=32, patch_size=16, depths=1, num_heads=4, embedding_dim=256, num_classes=1000).eval().cuda()
    x = torch.randn(64, 3, 32, 32, device="cuda")
    with torch.no_grad():
        tags = [t for t, *_ in mi355attn.kernel_trace(lambda: m(x))]
    assert any("gemm_small_kernel" in t and "N=1000" in t for t in tags), tags


@pytest.mark.parametrize("prec", [1, 2])
def test_fused_mlp_four_tile_variant_is_bit_identical(prec):
    """Option "mlp_tt4" (8 waves x 4 token tiles at 256 registers; VERDICT rounds 2-4: 'the 256-VGPR MLP kernel with four token tiles per
    wave') changes which wave owns which tokens, not the arithmetic: same bits as the 16 x 2 default, with and without the projection in
    front, on a token count that leaves ragged chunks."""
    import mi355attn
    from mi355attn.modules import CSWinBlock
    torch.manual_seed(21)
    m = CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True, precision=prec).eval().cuda()
    x = torch.randn(3, 3136, 64, device="cuda")
    old = mi355attn.get_option("mlp_tt4")
    try:
        with torch.no_grad():
            mi355attn.set_option("mlp_tt4", 0)
            y0 = m(x)
            mi355attn.set_option("mlp_tt4", 1)
            y1 = m(x)
    finally:
        mi355attn.set_option("mlp_tt4", old)
    assert torch.isfinite(y0).all() and torch.equal(y0, y1)


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,C", [(8, 512), (5, 512), (16, 256), (3, 768)])
def test_mixer_statistics_inside_the_token_kernel_are_bit_identical(B, C, prec):
    """Option "mixer_stats": LayerNorm row statistics computed inside mixer_token_kernel (phase 0, every workgroup of an image reads the
    image's rows once more; the two workgroups of an image get block ids 8 apart when B % 8 == 0) against the row_stats_kernel pre-pass:
    same per-lane sums in the same order, hence the same bits -- for one / two / three workgroups per image and both id mappings."""
    import mi355attn
    from mi355attn.modules import MixerLayer
    torch.manual_seed(B * 100 + C)
    m = MixerLayer(C, 196, precision=prec).eval().cuda()
    with torch.no_grad():
        m.norm1.weight.uniform_(0.5, 1.5)
        m.norm1.bias.normal_(0, 0.2)
    x = torch.randn(B, 196, C, device="cuda") * 1.7 + 0.3
    old = mi355attn.get_option("mixer_stats")
    try:
        with torch.no_grad():
            mi355attn.set_option("mixer_stats", 0)
            t0 = [t for t, *_ in mi355attn.kernel_trace(lambda: m(x))]
            y0 = m(x)
            mi355attn.set_option("mixer_stats", 1)
            seen = []
            t1 = [t for t, *_ in mi355attn.kernel_trace(lambda: seen.append(m(x)))]
    finally:
        mi355attn.set_option("mixer_stats", old)
    assert any("row_stats_kernel" in t for t in t0), t0
    if C == 512:                                                       # phase 0 is built for C = 512 (two float4 per lane and row)
        assert not any("row_stats_kernel" in t for t in t1) and any("mixer_token_kernel<stats>" in t for t in t1), t1
    assert torch.isfinite(y0).all() and torch.equal(y0, seen[0])


@pytest.mark.parametrize("prec", [1, 2])
def test_attention_core_on_seven_waves_is_bit_identical(prec):
    """Option "attn_nw" = 7: the 13 query tiles of a 197-token head on seven waves (six carry two tiles, one carries one) instead of eight
    (five carry two, three carry one).  A query tile's arithmetic does not depend on the wave that owns it: same bits."""
    import mi355attn
    from mi355attn.modules import Attention
    torch.manual_seed(3)
    m = Attention(768, 12, precision=prec).eval().cuda()
    x = torch.randn(5, 197, 768, device="cuda")
    old = mi355attn.get_option("attn_nw")
    try:
        with torch.no_grad():
            mi355attn.set_option("attn_nw", 8)
            y8 = m(x)
            mi355attn.set_option("attn_nw", 7)
            y7 = m(x)
    finally:
        mi355attn.set_option("attn_nw", old)
    assert torch.isfinite(y8).all() and torch.equal(y8, y7)


# ---- gemm16_w4.hip: the one-wave-per-SIMD persistent kernel --------------------------------------------------------------------------
W4_CASES = [  # M, N, K, gelu, bias
    (2048, 2304, 768, False, True),        # fewer tiles than CUs (72)
    (256 * 20, 768, 768, True, True),      # 60 tiles, GELU epilogue
    (256 * 90, 768, 320, False, False),    # 270 tiles = 1 round + 14, odd number of K-tiles (5), no bias
    (256 * 30, 512, 128, False, True),     # two K-tiles: the shortest stream the kernel takes
    (256 * 33, 1024, 1024, True, False),   # 132 tiles, GELU, no bias
    (256 * 131, 512, 576, False, True),    # 262 tiles = 1 round + 6
    (256 * 197, 2304, 768, False, True),   # the qkv product of ViT-Base at the timed size (B = 256): 1773 tiles = 6.93 rounds
    (256 * 197, 3072, 768, True, True),    # fc1 of ViT-Base at the timed size, GELU epilogue: 2364 tiles
]


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("case", W4_CASES)
def test_one_wave_per_simd_gemm_is_bit_identical_to_the_eight_wave_kernel(case, prec):
    """gemm16_w4.hip (round 5; 16-bit outputs): hand-placed MFMA / ds_read / LDS-DMA stream, accumulators in AGPRs, bias through an LDS-DMA
    piece, two-slab pipelined epilogue -- same K order per output as gemm16_p8.hip / gemm16_pa.hip, so the bits must agree (variant 17
    against 15 / 16 / the default dispatch), run to run as well.  fp32 outputs are refused (they belong to gemm16_pa)."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K, gelu, bias = case
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = torch.float16 if prec == 1 else torch.bfloat16
    x16 = torch.randn(M, K, generator=g).to(dev).to(dt)
    w16 = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dt)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    act = F.ACT_GELU if gelu else F.ACT_NONE
    old = mi355attn.get_option("gemm_variant")
    outs = {}
    try:
        for v in (17, 0, 15, 16, 17):
            mi355attn.set_option("gemm_variant", v)
            try:
                y = F.linear16(x16, w16, b, act=act, out16=True, precision=prec)
            except mi355attn.Mi355Error:
                assert v in (15, 16), "variant %d refused %r" % (v, case)        # the forced partners may not take every shape
                continue
            torch.cuda.synchronize()
            outs.setdefault(v, []).append(y.clone())
    finally:
        mi355attn.set_option("gemm_variant", old)
    _drain_range()
    assert len(outs[17]) == 2 and torch.equal(outs[17][0], outs[17][1])
    mi355attn.set_option("gemm_variant", 17)
    try:
        with pytest.raises(mi355attn.Mi355Error):
            F.linear16(x16, w16, b, act=act, out16=False, precision=prec)
    finally:
        mi355attn.set_option("gemm_variant", old)
    for v, ys in outs.items():
        assert torch.equal(ys[0], outs[17][0]), "variant %d differs from gemm16_w4 on %r" % (v, case)
    ref = x16.float() @ w16.float().t() + (b if b is not None else 0.0)
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    assert rel_fro(outs[17][0].float(), ref) < (3e-3 if prec == 1 else 8e-3)


def test_qkv_shaped_products_take_the_one_wave_per_simd_kernel():
    """Dispatch: 16-bit outputs, whole 256 x 256 tiles, 576 <= K < 1536 -> gemm16_w4 (option "gemm_w4"); the same call with the option off runs
    gemm16_p8; the results agree bit for bit."""
    import mi355attn
    from mi355attn import functional as F
    dev = torch.device("cuda", 0)
    x16 = torch.randn(256 * 300, 768, device=dev).half()
    w16 = (torch.randn(2304, 768, device=dev) / 27.7).half()
    b = torch.randn(2304, device=dev)
    old = mi355attn.get_option("gemm_w4")
    res = {}
    try:
        for v in (1, 0):
            mi355attn.set_option("gemm_w4", v)
            box = {}
            def run():
                box["y"] = F.linear16(x16, w16, b, out16=True, precision=1)
            tags = [t for t, *_ in mi355attn.kernel_trace(run)]
            torch.cuda.synchronize()
            res[v] = (box["y"].clone(), tags)
    finally:
        mi355attn.set_option("gemm_w4", old)
    assert any("gemm16_w4_kernel" in t for t in res[1][1]), res[1][1]
    assert any("gemm16_p8_kernel" in t for t in res[0][1]) and not any("gemm16_w4_kernel" in t for t in res[0][1]), res[0][1]
    assert torch.equal(res[0][0], res[1][0])


def test_one_wave_per_simd_gemm_under_graph_capture_and_on_a_side_stream():
    """gemm16_w4 has no inter-workgroup exchange, no workspace and no host-side state per launch: a captured launch replays on changed
    inputs bit for bit like an eager one, and a launch on a non-default stream gives the same bits."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K = 256 * 64, 2304, 768
    torch.manual_seed(5)
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    tags = [t for t, *_ in mi355attn.kernel_trace(lambda: F.linear16(x16, w16, b, out16=True, precision=1))]
    assert any("gemm16_w4_kernel" in t for t in tags), tags                  # 576 tiles >= one round: the dispatch takes the kernel
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y_cap = F.linear16(x16, w16, b, out16=True, precision=1)
    torch.cuda.current_stream().wait_stream(s)
    for rep in range(3):
        x16.copy_(torch.randn(M, K, device="cuda").half())
        g.replay()
        torch.cuda.synchronize()
        y_ref = F.linear16(x16, w16, b, out16=True, precision=1)
        torch.cuda.synchronize()
        assert torch.equal(y_cap, y_ref), "replay %d" % rep
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y_side = F.linear16(x16, w16, b, out16=True, precision=1)
    side.synchronize()
    assert torch.equal(y_side, y_ref)
    _drain_range()
