"""Extra bench.py workloads: BASELINE.json configs[2..4] plus the class-surface blocks without a BASELINE config.

Each builder returns dict(name, blocks, gather, dtype) like bench.workload_c2.  `work` is the ALGORITHMIC FLOP count per
call from SURVEY.md 8(d) (FLOP = 2*MAC; softmax / GELU / LayerNorm not counted), `bound` = "mfma" (divided by the dense
bf16/fp16 MFMA peak regardless of the precision mode in use).
"""
import torch

import oracle as O
from oracle import aten_seq as A

ATEN_NOTE = "ATen-operator-sequence restatement of the reference forward (oracle/aten_seq.py: the same fused layer_norm / linear / gelu / softmax / batched matmul calls in the same order)"


def _seeded(ctor, seed=1234):
    torch.manual_seed(seed)
    return ctor().eval()


def _sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def workload_c3(B, dev):
    from mi355attn.modules import Attention
    m = _seeded(lambda: Attention(768, 12))
    torch.manual_seed(4321)
    x = torch.randn(B, 197, 768, device=dev)
    sd = _sd(m)
    blocks = [dict(name="ViT Attention(768,h12)", key="ViTAttn", module=m.to(dev), x=x, bound="mfma", work=1.048784e9 * B, cpu_n=32,
                   alt_bytes=2.0 * 197 * 768 * 4 * B + 4 * 768 * 768 * 2,      # x in + y out (fp32) + the 16-bit weights (319 MB)
                   cpu=lambda xs: A.vit_attention_aten(xs, sd, 12), cpu_note=ATEN_NOTE)]
    return dict(name="ViT-Base Attention fwd, x=(%d,197,768) (BASELINE configs[2])" % B, blocks=blocks, dtype="f16")


def workload_c4(B, dev):
    from mi355attn.modules import CSWinBlock, XCA, XCABlock
    cfgs = [("CSWinBlock s1 (64,56,h2,sp1)", (64, 56, 2), dict(split_size=1, qkv_bias=True), (3136, 64), 356.9e6,
             lambda sd: (lambda xs: O.cswin_block_forward_aten(xs, sd, 56, 2, 1))),
            ("CSWinBlock s2 (128,28,h4,sp2)", (128, 28, 4), dict(split_size=2, qkv_bias=True), (784, 128), 332.6e6,
             lambda sd: (lambda xs: O.cswin_block_forward_aten(xs, sd, 28, 4, 2))),
            ("CSWinBlock s3 (256,14,h8,sp7)", (256, 14, 8), dict(split_size=7, qkv_bias=True), (196, 256), 328.9e6,
             lambda sd: (lambda xs: O.cswin_block_forward_aten(xs, sd, 14, 8, 7))),
            ("CSWinBlock s4 (512,7,h16,sp7,last)", (512, 7, 16), dict(split_size=7, qkv_bias=True, last_stage=True), (49, 512),
             313.65e6, lambda sd: (lambda xs: O.cswin_block_forward_aten(xs, sd, 7, 16, 7, True)))]
    blocks = []
    for name, args, kw, shp, flop, mk in cfgs:
        m = _seeded(lambda: CSWinBlock(*args, **kw))
        torch.manual_seed(4321)
        x = torch.randn(B, *shp, device=dev)
        blocks.append(dict(name=name, key="CSWin_" + name.split()[1], module=m.to(dev), x=x, bound="mfma", work=flop * B, cpu=mk(_sd(m)), cpu_n=32,
                           alt_bytes=2.0 * shp[0] * shp[1] * 4 * B,      # both roofs: x in + y out, the block's activation bytes
                           cpu_note="ATen-sequence restatement (strided window views, bmm, softmax, grouped conv2d, fused layer_norm / linear / "
                                    "gelu): the operator sequence of cswin.py:101-127,176-197; within 0.9-1.4x of the real reference on the "
                                    "build container's CPU"))
    xb = _seeded(lambda: XCABlock(384, 8, qkv_bias=True, eta=1.0))
    xa = _seeded(lambda: XCA(384, 8, qkv_bias=True))
    torch.manual_seed(4321)
    x = torch.randn(B, 196, 384, device=dev)
    sdb, sda = _sd(xb), _sd(xa)
    blocks.append(dict(name="XCABlock(384,h8)", key="XCABlock", module=xb.to(dev), x=x, fwd_args=(14, 14), bound="mfma", work=710.7e6 * B,
                       alt_bytes=2.0 * 196 * 384 * 4 * B, cpu_n=32, cpu=lambda xs: A.xca_block_aten(xs, sdb, 8, 14, 14), cpu_note=ATEN_NOTE))
    blocks.append(dict(name="XCA(384,h8)", key="XCA", module=xa.to(dev), x=x, bound="mfma", work=(173.4 + 7.2 + 7.2 + 57.8) * 1e6 * B,
                       alt_bytes=2.0 * 196 * 384 * 4 * B, cpu_n=32, cpu=lambda xs: A.xca_aten(xs, sda, 8), cpu_note=ATEN_NOTE))
    return dict(name="CSWin-T blocks s1-s4 + XCiT-S XCABlock/XCA fwd, B=%d (BASELINE configs[3])" % B, blocks=blocks,
                dtype="f16")


def workload_c5(B, dev):
    from mi355attn.modules import VisionTransformer
    m = _seeded(lambda: VisionTransformer(num_heads=12))
    torch.manual_seed(4321)
    x = torch.randn(B, 3, 224, 224, device=dev)
    sd = _sd(m)

    # gather=True: bench.py all-gathers this block's logits (mi355attn.dist.gather_batch: one RCCL all-gather over xGMI, 1 MB per rank)
    blocks = [dict(name="VisionTransformer(ViT-Base/16, h12)", key="ViTBase", module=m.to(dev), x=x, bound="mfma", work=35.127656e9 * B,
                   # HBM side: images in + logits out + one pass over the 16-bit weights (0.33 GB at B = 256); the forward's
                   # every-tensor-materialised-once figure is ~30 GB, the counters say what really moved
                   alt_bytes=(3 * 224 * 224 + 1000) * 4.0 * B + 86.54e6 * 2,
                   cpu=lambda xs: A.vit_aten(xs, sd, 12, 12), cpu_n=64, cpu_budget_x=6.0, gather=True, cpu_note=ATEN_NOTE)]
    return dict(name="ViT-Base full fwd, %d images per GPU, logits all-gathered (BASELINE configs[4])" % B, blocks=blocks,
                gather=True, dtype="f16")


def workload_mixer(B, dev):
    from mi355attn.modules import MixerLayer
    m = _seeded(lambda: MixerLayer(512, 196))
    torch.manual_seed(4321)
    x = torch.randn(B, 196, 512, device=dev)
    sd = _sd(m)
    blocks = [dict(name="MixerLayer(512,196)", key="Mixer", module=m.to(dev), x=x, bound="mfma", work=924.8e6 * B, cpu_n=32,
                   alt_bytes=2.0 * 196 * 512 * 4 * B,
                   cpu=lambda xs: A.mixer_layer_aten(xs, sd), cpu_note=ATEN_NOTE)]
    return dict(name="MLP-Mixer layer fwd, x=(%d,196,512)" % B, blocks=blocks, dtype="f16")


def workload_da(B, dev):
    from mi355attn.modules import DoubleAttention
    blocks = []
    for C, cm, cn, hw, flop in ((64, 32, 32, 32, 21.0e6), (256, 128, 128, 56, 0.0)):
        m = _seeded(lambda: DoubleAttention(C, cm, cn))
        torch.manual_seed(4321)
        x = torch.randn(B, C, hw, hw, device=dev)
        sd = _sd(m)
        if not flop:
            n = hw * hw
            flop = 2.0 * n * ((cm + 2 * cn) * C + cm * cn * 2 + C * cm)
        # DoubleAttention is a mixed block -- graded on the HBM roofline (algorithmic bytes = read x + write y), FLOP rate
        # reported next to it (alt_*)
        blocks.append(dict(name="DoubleAttention(%d,%d,%d)@%dx%d" % (C, cm, cn, hw, hw), key="DA%d" % C, module=m.to(dev), x=x, bound="hbm",
                           work=2.0 * C * hw * hw * 4 * B, alt_work=flop * B, cpu_n=32 if C > 64 else 64,
                           cpu=(lambda sd_: (lambda xs: A.double_attention_aten(xs, sd_)))(sd), cpu_note=ATEN_NOTE))
    return dict(name="DoubleAttention fwd, B=%d" % B, blocks=blocks, dtype="f16")


def _full_model(ctor, title, flop_per_image, cpu_fn, B, dev):
    m = _seeded(ctor)
    torch.manual_seed(4321)
    x = torch.randn(B, 3, 224, 224, device=dev)
    sd = _sd(m)

    blocks = [dict(name=title, key="".join(c for c in title.split("(")[0] if c.isalnum()), module=m.to(dev), x=x, bound="mfma",
                   work=flop_per_image * B, cpu=lambda xs: cpu_fn(xs, sd), cpu_n=32, gather=True)]
    return dict(name="%s full fwd, %d images per GPU, logits all-gathered" % (title, B), blocks=blocks, gather=True, dtype="f16")


def workload_cswin(B, dev):
    """CSWin_64_12211_tiny_224 (cswin.py:360-363): stem conv7/4 + [1, 2, 21, 1] blocks + three conv3/2 merges + head."""
    from mi355attn.modules import CSWin_64_12211_tiny_224
    flop = (2.0 * 3136 * 147 * 64 + 356.9e6 + 2.0 * 784 * 576 * 128 + 2 * 332.6e6 + 2.0 * 196 * 1152 * 256 + 21 * 328.9e6
            + 2.0 * 49 * 2304 * 512 + 313.65e6 + 2.0 * 512 * 1000)
    return _full_model(CSWin_64_12211_tiny_224, "CSWin-T/224", flop, lambda xs, sd: O.cswin_forward(xs, sd), B, dev)


def workload_mixer_full(B, dev):
    """MLP_Mixer(dim 512, depth 12, patch 16) (mlp_mixer.py:53-79)."""
    from mi355attn.modules import MLP_Mixer
    flop = 2.0 * 196 * 768 * 512 + 12 * 924.8e6 + 2.0 * 512 * 1000
    return _full_model(MLP_Mixer, "MLP-Mixer(512, depth 12)", flop, lambda xs, sd: O.mixer_forward(xs, sd), B, dev)


def workload_xcit(B, dev):
    """xcit_nano_12_p16 (xcit.py:416-420): conv patch embedding + Fourier positions, 12 XCABlocks (dim 128, 4 heads), 2 class-attention
    blocks, head.  FLOPs per image: convs 97.5e6 + 12 x 81.2e6 (qkv, covariance core, proj, LPI, MLP) + 2 x 13.5e6 + head."""
    from mi355attn.modules import xcit_nano_12_p16
    flop = 97.5e6 + 12 * 81.2e6 + 2 * 13.5e6 + 2.0 * 128 * 1000
    return _full_model(xcit_nano_12_p16, "XCiT-nano-12/16", flop, lambda xs, sd: O.xcit_forward(xs, sd), B, dev)


def workload_zoo(B, dev):
    """SimAM, SRM, Gaussian GCT, LCT, GCT at the C2 shape: HBM-bound, algorithmic bytes = read x + write y."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
    from cases import perturb_all
    from mi355attn.modules import GCT, LCT, SRM, GaussianGCT, simam_module
    C, H = 256, 56
    torch.manual_seed(4321)
    x = torch.randn(B, C, H, H, device=dev)
    nbytes = 2.0 * C * H * H * 4
    blocks = []
    for name, ctor in (("simam_module", lambda: simam_module()), ("SRM(256)", lambda: SRM(C)), ("GaussianGCT(256)", lambda: GaussianGCT(C)),
                       ("LCT(256,16)", lambda: LCT(C, 16)), ("GCT(256,l2)", lambda: GCT(C))):
        m = _seeded(ctor)
        perturb_all(m)
        sd = _sd(m)
        if name.startswith("simam"):
            cpu = lambda xs: O.simam_forward(xs)
        elif name.startswith("SRM"):
            cpu = (lambda s: (lambda xs: O.srm_forward(xs, s["cfc.weight"], s["bn.weight"], s["bn.bias"], s["bn.running_mean"], s["bn.running_var"])))(sd)
        elif name.startswith("Gaussian"):
            cpu = lambda xs: O.gct_gauss_forward(xs)
        elif name.startswith("LCT"):
            cpu = (lambda s: (lambda xs: O.lct_forward(xs, s["w"], s["b"], 16)))(sd)
        else:
            cpu = (lambda s: (lambda xs: O.gct_forward(xs, s["alpha"], s["gamma"], s["beta"])))(sd)
        blocks.append(dict(name=name, key=name.split("(")[0], module=m.to(dev), x=x, bound="hbm", work=nbytes * B, cpu=cpu))
    return dict(name="SimAM+SRM+GaussianGCT+LCT+GCT fwd, x=(%d,256,56,56) fp32 per GPU" % B, blocks=blocks, dtype="f32")


def workload_zoo2(B, dev):
    """GCModule, CoordinateAttention, TripletAttention, BAM, SKLayer, CAM (second group) at the C2 shape.  The gates are
    HBM-bound (algorithmic bytes = read x + write y); SKLayer's grouped 3x3 branches and CAM's Gram products are counted in FLOPs."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
    from cases import perturb_all
    from mi355attn.modules import BAM, CAM, CoordinateAttention, GCModule, SKLayer, TripletAttention
    C, H = 256, 56
    torch.manual_seed(4321)
    x = torch.randn(B, C, H, H, device=dev)
    nbytes = 2.0 * C * H * H * 4
    sk_flop = 2.0 * 2 * H * H * C * (C // 32) * 9                  # two grouped 3x3 branches
    cam_flop = 2.0 * 2 * C * C * H * H                             # X X^T and attn X
    blocks = []
    for name, ctor, orc, bound, work in (
            ("GCModule(256)", lambda: GCModule(C), lambda xs, s: O.gc_forward(xs, s), "hbm", nbytes),
            ("CoordinateAttention(256)", lambda: CoordinateAttention(C, C), lambda xs, s: O.coordatt_forward(xs, s), "hbm", nbytes),
            ("TripletAttention(7)", lambda: TripletAttention(), lambda xs, s: O.triplet_forward(xs, s), "hbm", nbytes),
            ("BAM(256)", lambda: BAM(C), lambda xs, s: O.bam_forward(xs, s), "hbm", nbytes),
            ("SKLayer(256,256)", lambda: SKLayer(C, C), lambda xs, s: O.sk_forward(xs, s), "mfma", sk_flop),
            ("CAM", lambda: CAM(), lambda xs, s: O.cam_forward(xs, s), "mfma", cam_flop)):
        m = _seeded(ctor)
        perturb_all(m)
        m.eval()
        sd = _sd(m)
        cpu = (lambda f, s: (lambda xs: f(xs, s)))(orc, sd)
        blocks.append(dict(name=name, key=name.split("(")[0], module=m.to(dev), x=x, bound=bound, work=work * B, cpu=cpu))
    return dict(name="GC+CoordAtt+Triplet+BAM+SK+CAM fwd, x=(%d,256,56,56) fp32 per GPU" % B, blocks=blocks, dtype="f32")


def workload_f1(B, dev):
    """SURVEY 8 f1 (the plain-MHSA copies of the other ViT files) at their NATIVE stage shapes, B images per GPU: SETR (dim 256, 4 heads,
    1024 tokens), PVT stage 1 / stage 3 (spatial-reduction K/V: sr 8 at 56 x 56, sr 2 at 14 x 14), CMT (sr 2 + relative position bias at
    28 x 28), SegFormer (sr 8, fused kv projection), P2T (pooled-pyramid K/V at 56 x 56).  Modules, weights and forward arguments are the
    golden cases of tests/golden/cases.py (built through the drop-in import paths); FLOPs = projections + QK^T + PV (2 x MAC); the
    spatial-reduction convolutions are counted on their reduced token grid."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
    from cases import BY_ID, PREP, make_arg

    def mhsa_flop(N, Nkv, C, kv_proj=True, sr_conv=0.0):
        return 2.0 * N * C * C * 2 + (2.0 * Nkv * C * 2 * C if kv_proj else 0.0) + 2.0 * 2 * N * Nkv * C + sr_conv

    specs = [("setr_attn", "SETR_Attn", mhsa_flop(1024, 1024, 256)),
             ("pvt_attn_s1", "PVT_s1", mhsa_flop(3136, 49, 64, sr_conv=2.0 * 49 * 64 * 64 * 64)),
             ("pvt_attn_s3", "PVT_s3", mhsa_flop(196, 49, 320, sr_conv=2.0 * 49 * 320 * 320 * 4)),
             ("cmt_attn", "CMT", mhsa_flop(784, 196, 128, sr_conv=2.0 * 196 * 128 * 4)),
             ("segformer_attn", "SegFormer", mhsa_flop(3136, 49, 64, sr_conv=2.0 * 49 * 64 * 64 * 64)),
             ("p2t_attn", "P2T", mhsa_flop(3136, 16 + 9 + 4 + 4, 64))]
    blocks = []
    for cid, key, flop in specs:
        c = BY_ID[cid]
        cls = getattr(importlib.import_module(c["mod"]), c["cls"])
        m = _seeded(lambda: cls(*c.get("args", ()), **c.get("kwargs", {})))
        if c.get("prep"):
            PREP[c["prep"]](m)
            m.eval()
        torch.manual_seed(4321)
        x = torch.randn(B, *c["shape"][1:], device=dev)
        sd = _sd(m)
        args = [make_arg(a) for a in c.get("fwd_args", ())]
        dargs = tuple(a.to(dev) if isinstance(a, (torch.Tensor, torch.nn.Module)) else a for a in args)
        blocks.append(dict(name="%s %s%s @ %s" % (c["mod"].split(".")[-1], c["cls"], c.get("args", ()), "x".join(str(v) for v in c["shape"][1:])),
                           key=key, module=m.to(dev), x=x, fwd_args=dargs, bound="mfma", work=flop * B, cpu_n=32,
                           alt_bytes=2.0 * x[0].numel() * 4 * B,
                           cpu=(lambda cc, s: (lambda xs: cc["oracle"](xs, s, torch.float32)))(c, sd)))
    return dict(name="f1: plain-MHSA copies (SETR, PVT s1/s3, CMT, SegFormer, P2T) at native shapes, B=%d" % B, blocks=blocks, dtype="f16")


WORKLOADS = {"f1": workload_f1, "zoo2": workload_zoo2, "zoo": workload_zoo, "xcit": workload_xcit, "cswin": workload_cswin, "mixer_full": workload_mixer_full, "c3": workload_c3, "c4": workload_c4, "c5": workload_c5, "mixer": workload_mixer, "da": workload_da}
