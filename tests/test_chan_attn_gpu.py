"""GPU tests (-m gpu): edge cases and full-size properties of the SE / ECA / CBAM kernels."""
import pytest
import torch

import oracle as O
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _mods(C, red=16, ks=7):
    from mi355attn.modules import CBAM, ECALayer, SELayer
    torch.manual_seed(11)
    return SELayer(C, red).eval(), ECALayer(C).eval(), CBAM(C, red, ks).eval()


# ragged spatial sizes (HW % 4 != 0 -> scalar lanes), channel counts that are not multiples of the
# 16-row workgroup slab, single image, 1x1 maps, wide maps that need several spatial-gate bands
SHAPES = [(1, 16, 1, 1), (2, 48, 7, 9), (3, 32, 5, 5), (2, 80, 13, 1), (1, 64, 3, 3), (2, 100, 10, 10),
          (1, 32, 64, 200), (5, 256, 14, 14), (2, 64, 32, 32), (2, 24, 112, 112)]


@pytest.mark.parametrize("shape", SHAPES)
def test_edge_shapes(shape):
    B, C, H, W = shape
    red = 16 if C >= 32 else 4
    se, eca, cbam = _mods(C, red, 7 if min(H, W) >= 3 else 3)
    torch.manual_seed(5)
    x = torch.randn(*shape)
    xd = x.cuda()
    with torch.no_grad():
        y_se = se.cuda()(xd).cpu()
        y_eca = eca.cuda()(xd).cpu()
        y_cbam = cbam.cuda()(xd).cpu()
        y_ca = cbam.ca(xd).cpu()
        y_sa = cbam.sa(xd).cpu()
    sd = {k: v.cpu() for k, v in cbam.state_dict().items()}
    assert_parity(y_se, O.se_forward(x, se.fc[0].weight, se.fc[2].weight), 1e-5, f"se{shape}")
    assert_parity(y_eca, O.eca_forward(x, eca.conv.weight), 1e-5, f"eca{shape}")
    assert_parity(y_cbam, O.cbam_forward(x, sd["ca.fc.0.weight"], sd["ca.fc.2.weight"], sd["sa.conv.weight"]), 1e-5,
                  f"cbam{shape}")
    assert_parity(y_ca, O.cbam_channel_forward(x, sd["ca.fc.0.weight"], sd["ca.fc.2.weight"]), 1e-5, f"ca{shape}")
    assert_parity(y_sa, O.cbam_spatial_forward(x, sd["sa.conv.weight"]), 1e-5, f"sa{shape}")


ECA_SINGLE_SHAPES = [(2, 64, 32, 32), (3, 256, 56, 56), (1, 8, 4, 4), (2, 16, 2, 2), (2, 1024, 14, 14), (1, 24, 64, 64), (5, 40, 12, 12),
                     (1, 4096, 8, 8)]


@pytest.mark.parametrize("shape", ECA_SINGLE_SHAPES)
def test_eca_single_read_kernel(shape):
    """ECA with x read once (halo rows re-summed per workgroup) vs the oracle and vs the two-pass path; k = 3, 5 and 7 (C = 4096),
    first / last channel groups exercise the zero padding, every NV instantiation is hit."""
    import mi355attn
    B, C, H, W = shape
    _, eca, _ = _mods(C, 4)
    torch.manual_seed(11)
    x = torch.randn(*shape)
    xd = x.cuda()
    old = mi355attn.get_option("eca_single")
    try:
        mi355attn.set_option("eca_single", 1)
        with torch.no_grad():
            y1 = eca.cuda()(xd).cpu()
            y1b = eca(xd).cpu()
        mi355attn.set_option("eca_single", 0)
        with torch.no_grad():
            y0 = eca(xd).cpu()
    finally:
        mi355attn.set_option("eca_single", old)
    assert torch.equal(y1, y1b)
    assert_parity(y1, O.eca_forward(x, eca.conv.weight.cpu()), 1e-5, f"eca single {shape}")
    assert_parity(y1, y0, 1e-6, "eca single read vs two pass")


@pytest.mark.parametrize("shape", ECA_SINGLE_SHAPES + [(70, 64, 28, 28), (3, 1024, 8, 8)])
def test_se_single_read_kernel(shape, monkeypatch):
    """SE with x read once (register-resident rows, {mean, tag} granule exchange between the workgroups of an image) vs the
    oracle and vs the two-pass path; the bounded-poll error word is checked after every call."""
    import mi355attn
    monkeypatch.setenv("MI355_CHECK_SYNC", "1")
    B, C, H, W = shape
    se, _, _ = _mods(C, 16 if C >= 32 else 4)
    torch.manual_seed(13)
    x = torch.randn(*shape)
    xd = x.cuda()
    try:
        mi355attn.set_option("se_single", 1)
        with torch.no_grad():
            y1 = se.cuda()(xd).cpu()
            y1b = se(xd).cpu()
        mi355attn.set_option("se_single", 0)
        with torch.no_grad():
            y0 = se(xd).cpu()
    finally:
        mi355attn.set_option("se_single", 1)
    assert torch.equal(y1, y1b)
    assert_parity(y1, O.se_forward(x, se.fc[0].weight.cpu(), se.fc[2].weight.cpu()), 1e-5, f"se single {shape}")
    assert_parity(y1, y0, 1e-6, "se single read vs two pass")


def test_se_single_granule_protocol_under_repetition(monkeypatch):
    """300 back-to-back launches reusing one workspace (granules re-zeroed by the memset node each time): every run must equal
    the first -- a stale or torn granule would change a gate -- and no poll may time out."""
    monkeypatch.setenv("MI355_CHECK_SYNC", "0")
    se, _, _ = _mods(256)
    torch.manual_seed(8)
    x = torch.randn(40, 256, 28, 28).cuda()
    with torch.no_grad():
        first = se.cuda()(x).clone()
        for _ in range(300):
            y = se(x)
        monkeypatch.setenv("MI355_CHECK_SYNC", "1")
        last = se(x)
    assert torch.equal(y, first) and torch.equal(last, first)
    assert_parity(first.cpu(), O.se_forward(x.cpu(), se.fc[0].weight.cpu(), se.fc[2].weight.cpu()), 1e-5, "se single repeat")


CBAM_SINGLE_SHAPES = [(2, 64, 32, 32), (3, 256, 56, 56), (1, 8, 4, 4), (2, 16, 2, 2), (2, 256, 14, 14), (2, 24, 112, 112), (5, 40, 12, 12),
                      (1, 512, 8, 8), (40, 64, 28, 28), (2, 100, 10, 10), (1, 32, 64, 128), (3, 48, 16, 20)]


@pytest.mark.parametrize("shape", CBAM_SINGLE_SHAPES)
def test_cbam_single_read_kernel(shape, monkeypatch):
    """CBAM with x read once (row bands of all channels in registers, three granule hops between the bands of an image) vs the
    oracle and vs the three-pass path.  Shapes cover both segment widths, all NV instantiations, channel counts that do not
    fill the last register slot, one band per image, many bands per image and kernel sizes 7 and 3."""
    import mi355attn
    monkeypatch.setenv("MI355_CHECK_SYNC", "1")
    B, C, H, W = shape
    ks = 7 if min(H, W) >= 4 else 3
    _, _, cbam = _mods(C, 16 if C >= 32 else 4, ks)
    torch.manual_seed(17)
    x = torch.randn(*shape)
    xd = x.cuda()
    try:
        mi355attn.set_option("cbam_single", 1)
        with torch.no_grad():
            y1 = cbam.cuda()(xd).cpu()
            y1b = cbam(xd).cpu()
        mi355attn.set_option("cbam_single", 0)
        with torch.no_grad():
            y0 = cbam(xd).cpu()
    finally:
        mi355attn.set_option("cbam_single", 1)
    sd = {k: v.cpu() for k, v in cbam.state_dict().items()}
    assert torch.equal(y1, y1b)
    assert_parity(y1, O.cbam_forward(x, sd["ca.fc.0.weight"], sd["ca.fc.2.weight"], sd["sa.conv.weight"]), 1e-5, f"cbam single {shape}")
    assert_parity(y1, y0, 2e-6, "cbam single read vs three pass")


def test_cbam_single_granule_protocol_under_repetition(monkeypatch):
    """200 back-to-back launches on one workspace: every run equals the first, no poll times out."""
    monkeypatch.setenv("MI355_CHECK_SYNC", "0")
    _, _, cbam = _mods(256)
    torch.manual_seed(8)
    x = torch.randn(24, 256, 28, 28).cuda()
    with torch.no_grad():
        first = cbam.cuda()(x).clone()
        for _ in range(200):
            y = cbam(x)
        monkeypatch.setenv("MI355_CHECK_SYNC", "1")
        last = cbam(x)
    assert torch.equal(y, first) and torch.equal(last, first)


def test_persistent_workspace_epochs(monkeypatch):
    """With "ws_persistent" (what the Python binding runs with) the exchange area is zeroed once and every launch carries a fresh
    tag.  Interleave shapes and ops that share nothing but the process-wide epoch counter, switch the option off and on again,
    and detour through the two-pass kernel on the same workspace: every result must stay bit-identical to the first."""
    import mi355attn
    monkeypatch.setenv("MI355_CHECK_SYNC", "1")
    assert mi355attn.get_option("ws_persistent") == 1
    se_a, _, cbam_a = _mods(64)
    se_b, _, cbam_b = _mods(256)
    torch.manual_seed(21)
    xa, xb = torch.randn(6, 64, 28, 28).cuda(), torch.randn(3, 256, 14, 14).cuda()
    mods = [(se_a.cuda(), xa), (cbam_a.cuda(), xa), (se_b.cuda(), xb), (cbam_b.cuda(), xb)]
    with torch.no_grad():
        first = [m(x).clone() for m, x in mods]
        for rnd in range(12):
            if rnd == 4:
                mi355attn.set_option("ws_persistent", 0)
            if rnd == 7:
                mi355attn.set_option("ws_persistent", 1)
            if rnd == 9:                                   # detour through the two-pass kernel on the same dedicated workspace
                mi355attn.set_option("se_single", 0)
                se_a(xa)
                mi355attn.set_option("se_single", 1)
            for (m, x), f in zip(mods, first):
                assert torch.equal(m(x), f), f"round {rnd}"
    mi355attn.set_option("ws_persistent", 1)


def test_eca_single_is_independent_of_batch_grouping():
    """mean(b,c) is accumulated in one fixed order whichever workgroup needs it, so an image's result does not depend on the batch
    it is part of (slice numbering / XCD placement change with B)."""
    _, eca, _ = _mods(256)
    torch.manual_seed(12)
    x = torch.randn(19, 256, 28, 28).cuda()
    with torch.no_grad():
        full = eca.cuda()(x)
        part = eca(x[7:12].contiguous())
    assert torch.equal(full[7:12], part)


def test_non_contiguous_and_offset_inputs():
    se, _, _ = _mods(64)
    torch.manual_seed(2)
    big = torch.randn(2, 64, 20, 24)
    view = big[:, :, 2:18, 4:20]                      # non-contiguous view -> wrapper makes it dense
    with torch.no_grad():
        y = se.cuda()(view.cuda()).cpu()
    assert_parity(y, O.se_forward(view.contiguous(), se.fc[0].weight, se.fc[2].weight), 1e-5, "se[view]")


def test_chunk_option_does_not_change_results():
    """The Infinity-Cache chunking / non-temporal / order knobs of the multi-pass kernels reorder launches and change cache hints only:
    outputs must be bit-identical (the single-read kernels are switched off so that the multi-pass path is what runs)."""
    import mi355attn
    se, eca, cbam = _mods(64)
    torch.manual_seed(9)
    x = torch.randn(13, 64, 28, 28).cuda()
    outs = []
    old = {k: mi355attn.get_option(k) for k in ("chunk_images", "nt", "reverse", "se_single", "eca_single", "cbam_single")}
    for k in ("se_single", "eca_single", "cbam_single"):
        mi355attn.set_option(k, 0)
    for chunk, nt, rev in ((0, 3, 0), (1, 0, 0), (5, 1, 1), (13, 2, 1), (4, 3, 1)):
        mi355attn.set_option("chunk_images", chunk)
        mi355attn.set_option("nt", nt)
        mi355attn.set_option("reverse", rev)
        with torch.no_grad():
            outs.append((se.cuda()(x).clone(), eca.cuda()(x).clone(), cbam.cuda()(x).clone()))
    for k, v in old.items():
        mi355attn.set_option(k, v)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("which", ["se", "eca", "cbam"])
def test_full_size_properties(which):
    """BASELINE config C2, x = (256,256,56,56): size-independent properties + sampled-image oracle checks.

    * per-row gate constancy (SE/ECA): y[b,c,:] / x[b,c,:] is one scalar in (0,1);
    * batch independence: the output of image b does not depend on the other images (run a 3-image
      sub-batch of images 0,127,255 and compare bit-for-bit with the same rows of the full run);
    * images 0, 127, 255 against the oracle;
    * run-to-run bit identity.
    """
    se, eca, cbam = _mods(256)
    m = {"se": se, "eca": eca, "cbam": cbam}[which].cuda()
    g = torch.Generator(device="cpu").manual_seed(4321)
    pick = [0, 127, 255]
    x = torch.empty(256, 256, 56, 56, device="cuda")
    host = {}
    for b0 in range(0, 256, 32):                       # fill in slabs to bound host memory
        blk = torch.randn(32, 256, 56, 56, generator=g)
        x[b0:b0 + 32] = blk.cuda()
        for b in pick:
            if b0 <= b < b0 + 32:
                host[b] = blk[b - b0].clone()
    with torch.no_grad():
        y = m(x)
        y2 = m(x)
        sub = m(x[pick].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(y, y2), "run-to-run results differ"
    assert torch.equal(y[pick], sub), "output of an image depends on its batch neighbours"   # same kernels, any batch
    assert torch.isfinite(y).all()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    xs = torch.stack([host[b] for b in pick])
    if which == "se":
        ref = O.se_forward(xs, sd["fc.0.weight"], sd["fc.2.weight"])
    elif which == "eca":
        ref = O.eca_forward(xs, sd["conv.weight"])
    else:
        ref = O.cbam_forward(xs, sd["ca.fc.0.weight"], sd["ca.fc.2.weight"], sd["sa.conv.weight"])
    assert_parity(y[pick].cpu(), ref, 1e-5, which + "[full-size sample]")
    if which in ("se", "eca"):
        ratio = (y[pick] / x[pick]).reshape(3 * 256, -1)
        ok = x[pick].reshape(3 * 256, -1).abs() > 1e-3
        lo = torch.where(ok, ratio, torch.full_like(ratio, 2.0)).amin(dim=1)
        hi = torch.where(ok, ratio, torch.full_like(ratio, -1.0)).amax(dim=1)
        assert float((hi - lo).max()) < 1e-4 and float(lo.min()) > 0.0 and float(hi.max()) < 1.0


# ---- SimAM / SRM / Gaussian GCT / LCT / GCT -------------------------------------------------------------------------
def _zoo_mods(C, groups):
    from mi355attn.modules import GCT, LCT, SRM, GaussianGCT, simam_module
    from cases import perturb_all
    torch.manual_seed(31)
    mods = dict(simam=simam_module(), srm=SRM(C).eval(), gctg=GaussianGCT(C), lct=LCT(C, groups), gct=GCT(C), gct_l1=GCT(C, mode="l1"),
                gct_l1r=GCT(C, mode="l1", after_relu=True))
    for m in mods.values():
        perturb_all(m)
    return mods


def _zoo_ref(name, m, x):
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    if name == "simam":
        return O.simam_forward(x, m.e_lambda)
    if name == "srm":
        return O.srm_forward(x, sd["cfc.weight"], sd["bn.weight"], sd["bn.bias"], sd["bn.running_mean"], sd["bn.running_var"], m.bn.eps)
    if name == "gctg":
        return O.gct_gauss_forward(x, m.c, m.eps)
    if name == "lct":
        return O.lct_forward(x, sd["w"], sd["b"], m.groups, m.eps)
    return O.gct_forward(x, sd["alpha"], sd["gamma"], sd["beta"], m.epsilon, m.mode, m.after_relu)


ZOO_SHAPES = [(2, 64, 32, 32, 8), (3, 256, 56, 56, 16), (2, 48, 7, 9, 4), (1, 20, 5, 5, 5), (2, 16, 2, 2, 2), (1, 24, 64, 64, 3),
              (5, 40, 12, 12, 8), (2, 12, 3, 1, 4), (1, 1024, 14, 14, 32)]


@pytest.mark.parametrize("single", [1, 0])
@pytest.mark.parametrize("shape", ZOO_SHAPES)
def test_zoo_gates_vs_oracle(shape, single, monkeypatch):
    """All five modules (GCT in its three variants) against the oracle: shapes that take the single-read register path, shapes that
    cannot (HW % 4 != 0, C % 8 != 0) and the forced two-pass path; run-to-run identical."""
    import mi355attn
    monkeypatch.setenv("MI355_CHECK_SYNC", "1")
    B, C, H, W, groups = shape
    mods = _zoo_mods(C, groups)
    torch.manual_seed(32)
    x = torch.randn(B, C, H, W)
    xd = x.cuda()
    mi355attn.set_option("zoo_single", single)
    try:
        for name, m in mods.items():
            with torch.no_grad():
                y = m.cuda()(xd)
                y2 = m(xd)
            assert torch.equal(y, y2), name
            assert_parity(y.cpu(), _zoo_ref(name, m, x), 2e-5, f"{name} {shape} single={single}")
    finally:
        mi355attn.set_option("zoo_single", 1)


def test_zoo_single_and_two_pass_agree_and_repeat():
    """Exchange protocol of the GCT / LCT kernels under repetition on one dedicated workspace, and agreement of the two paths."""
    import mi355attn
    mods = _zoo_mods(256, 16)
    torch.manual_seed(33)
    x = torch.randn(24, 256, 28, 28).cuda()
    for name in ("gctg", "lct", "gct", "gct_l1"):
        m = mods[name].cuda()
        with torch.no_grad():
            first = m(x).clone()
            for _ in range(100):
                y = m(x)
            assert torch.equal(y, first), name
            mi355attn.set_option("zoo_single", 0)
            try:
                two = m(x)
            finally:
                mi355attn.set_option("zoo_single", 1)
        assert_parity(first.cpu(), two.cpu(), 2e-6, f"{name}: single read vs two passes")


def test_exchange_kernels_under_graph_capture():
    """hipGraph capture of the channel-attention modules.  SE and CBAM record their single-read exchange kernels: the granule tag of a
    launch and the ticket word live in the workspace (epoch + 1; the last ticket draw of a launch resets the ticket and advances the
    epoch), so a replay is just another launch and eager calls may be interleaved with replays on the same workspace -- the results
    are the same bits either way.  GCT records its single-read kernel the same way since round 4 (tests/test_round4_gpu.py checks
    bit equality for GCT / LCT / Gaussian GCT); ECA has no exchange."""
    from mi355attn.modules import GCT
    se, _, cbam = _mods(64)
    gct = GCT(64)
    with torch.no_grad():
        gct.gamma.add_(0.5)
    _, eca, _ = _mods(64)
    mods = [se.cuda(), cbam.cuda(), gct.cuda(), eca.cuda()]
    torch.manual_seed(41)
    static_x = torch.randn(6, 64, 28, 28, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for m in mods:                                    # warm-up on the capture stream (allocations, dedicated workspaces)
            m(static_x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = [m(static_x) for m in mods]
    for rep in range(4):
        x = torch.randn(6, 64, 28, 28, device="cuda")
        static_x.copy_(x)
        g.replay()
        if rep == 2:
            g.replay()                                    # two replays back to back: the epoch advances inside the graph
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        with torch.no_grad():
            want = [m(x) for m in mods]                   # eager calls in between, same workspaces
        for a, b, m in zip(got, want, mods):
            assert_parity(a.cpu(), b.cpu(), 2e-6, f"replay {rep} {type(m).__name__}")
        assert torch.equal(got[0], want[0]), "SE: replay and eager launch differ"
        assert torch.equal(got[1], want[1]), "CBAM: replay and eager launch differ"
    import mi355attn
    mi355attn.sync_status(wait=True)


def test_exchange_kernels_under_graph_capture_at_the_bench_shape():
    """The same at the C2 geometry (256 channels, 56 x 56: 32 slices / 28 bands per image, several slices per workgroup): replays
    interleaved with eager launches on the same workspaces, bit-equal, and equal to the multi-pass kernels within fp32 noise."""
    import mi355attn
    se, _, cbam = _mods(256)
    mods = [se.cuda(), cbam.cuda()]
    torch.manual_seed(47)
    static_x = torch.randn(24, 256, 56, 56, device="cuda")
    with torch.no_grad():
        for m in mods:
            m(static_x)                                   # loads the code objects, makes the eager workspaces known
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = [m(static_x) for m in mods]
    for rep in range(3):
        x = torch.randn(24, 256, 56, 56, device="cuda")
        static_x.copy_(x)
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        with torch.no_grad():
            want = [m(x) for m in mods]
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), f"replay {rep}"
    mi355attn.set_option("se_single", 0); mi355attn.set_option("cbam_single", 0)
    try:
        with torch.no_grad():
            multi = [m(x) for m in mods]
    finally:
        mi355attn.set_option("se_single", 1); mi355attn.set_option("cbam_single", 1)
    assert_parity(got[0].cpu(), multi[0].cpu(), 2e-6, "SE single-read (replayed) vs two-pass")
    assert_parity(got[1].cpu(), multi[1].cpu(), 2e-6, "CBAM single-read (replayed) vs three-pass")
    mi355attn.sync_status(wait=True)


def test_exchange_kernels_captured_on_a_cold_workspace():
    """Capture WITHOUT a warm-up call: the workspace of the module is unknown to the library, so the zeroing of its exchange area is
    recorded with the launch and repeated by every replay (epoch 0 -> tag 1 each time); results must match eager launches and the
    oracle, before and after eager calls have made the workspace known."""
    se, _, cbam = _mods(64)
    mods = [se.cuda(), cbam.cuda()]
    torch.manual_seed(43)
    with torch.no_grad():                                 # code objects are loaded on first launch, which HIP forbids inside a capture:
        for m in mods:                                    # launch the kernels once on ANOTHER shape (its own workspace)
            m(torch.randn(2, 64, 28, 28, device="cuda"))
    torch.cuda.synchronize()
    static_x = torch.randn(5, 64, 28, 28, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = [m(static_x) for m in mods]
    for rep in range(3):
        x = torch.randn(5, 64, 28, 28, device="cuda")
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        with torch.no_grad():
            want = [m(x) for m in mods]
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), f"replay {rep}"
    ref = O.se_forward(x.cpu(), se.fc[0].weight.cpu(), se.fc[2].weight.cpu())
    assert_parity(got[0].cpu(), ref, 1e-5, "SE replay vs oracle")
    import mi355attn
    mi355attn.sync_status(wait=True)


@pytest.mark.parametrize("single", [1, 0])
@pytest.mark.parametrize("shape", [(2, 96, 28, 28), (3, 40, 7, 9), (2, 256, 56, 56)])
def test_se_variants_bias_and_hard_sigmoid(shape, single):
    """SE with excitation biases (efficientnet / mnasnet) and with the hard-sigmoid gate (ghostnet) through the single-read and the
    two-pass kernels, against the oracle."""
    import mi355attn
    from mi355attn.modules import SELayerBias, SqueezeExcite
    from cases import perturb_all
    B, C, H, W = shape
    torch.manual_seed(51)
    a, g = SELayerBias(C, 4).eval(), SqueezeExcite(C).eval()
    perturb_all(a); perturb_all(g)
    x = torch.randn(*shape)
    mi355attn.set_option("se_single", single)
    try:
        with torch.no_grad():
            ya, yg = a.cuda()(x.cuda()).cpu(), g.cuda()(x.cuda()).cpu()
    finally:
        mi355attn.set_option("se_single", 1)
    sa, sg = {k: v.cpu() for k, v in a.state_dict().items()}, {k: v.cpu() for k, v in g.state_dict().items()}
    assert_parity(ya, O.se_ex_forward(x, sa["fc.0.weight"], sa["fc.0.bias"], sa["fc.2.weight"], sa["fc.2.bias"]), 1e-5, "se + bias")
    assert_parity(yg, O.se_ex_forward(x, sg["conv_reduce.weight"], sg["conv_reduce.bias"], sg["conv_expand.weight"], sg["conv_expand.bias"],
                                      "hard_sigmoid"), 1e-5, "se + bias + hard sigmoid")
