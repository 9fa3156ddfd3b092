"""Functional layer: tensor-in / tensor-out wrappers over the C ABI (one per entry point of mi355attn.h).

Every function validates dtype/device, makes inputs contiguous, allocates the output (and borrows the
per-stream workspace) with torch -- device memory and streams are torch's job, the arithmetic is the
library's -- and raises ``Mi355Error`` on any non-zero return code.
"""
import ctypes
import weakref

import torch

from . import _ffi
from ._ffi import check, dptr, lib, require_device_f32, stream_ptr, workspace

PREC_STRICT, PREC_FP16, PREC_BF16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2     # ACT_RELU: fp32-in engine only (linear, conv2d_tokens)

_default_precision = PREC_FP16


def set_default_precision(p):
    """Process-wide default MFMA operand precision for modules that do not pin one (0 strict / 1 fp16 / 2 bf16)."""
    global _default_precision
    if p not in (PREC_STRICT, PREC_FP16, PREC_BF16):
        raise ValueError("precision must be 0 (strict), 1 (fp16) or 2 (bf16)")
    _default_precision = p


def default_precision():
    return _default_precision


def _prec(p):
    return _default_precision if p is None else p


def _opt(t, name):
    return None if t is None else require_device_f32(t, name)


# ---- channel / spatial attention -------------------------------------------------------------------------
def sync_status(wait=False):
    """Raise Mi355Error if an exchange kernel (single-read SE / CBAM / GCT / LCT) that has ALREADY executed ran out of its poll
    budget (include/mi355attn.h mi355_sync_status: a pinned host word, read without a device synchronisation).  `wait=True`
    synchronises the device first, so that the check covers every launch issued so far."""
    if wait:
        torch.cuda.synchronize()
    check(lib().mi355_sync_status(), "mi355_sync_status")


def range_status(wait=False):
    """Raise Mi355RangeError if a 16-bit producer (cast16, layernorm16, a GEMM epilogue with 16-bit output) that has ALREADY executed
    converted a finite value of magnitude >= 65520 to fp16 -- the tensor then holds inf where the fp32 reference is finite
    (mi355_range_status: a pinned host word, read without a device synchronisation).  `wait=True` synchronises first, so that the check
    covers every launch issued so far.  bf16 and strict mode cannot overflow this way."""
    if wait:
        torch.cuda.synchronize()
    code = lib().mi355_range_status()
    if code != 0:
        msg = lib().mi355_last_error()
        raise _ffi.Mi355RangeError(f"fp16 range guard (code {code}): {msg.decode() if msg else '?'}")


def _range_check():
    """Before every launch of the 16-bit dataflow: report an overflow of any EARLIER launch now instead of computing on inf.
    MI355_CHECK_RANGE=1 waits for the device first (debugging aid: the check then covers everything issued so far)."""
    import os
    range_status(wait=os.environ.get("MI355_CHECK_RANGE") == "1")


def _tensors_of(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors_of(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_of(o)


def _all_finite(obj):
    return all(bool(torch.isfinite(t).all()) for t in _tensors_of(obj) if t.is_floating_point())


class _forced_strict:
    """Strict mode for a re-run: the package default AND every sub-module that was built with an explicit 16-bit `precision=`
    (an explicit setting beats the default, so switching the default alone would recompute those modules in fp16)."""

    def __init__(self, module):
        self.module, self.saved, self.old = module, [], None

    def __enter__(self):
        self.old = default_precision()
        set_default_precision(PREC_STRICT)
        mods = self.module.modules() if isinstance(self.module, torch.nn.Module) else ()
        for m in mods:
            p = m.__dict__.get("precision", None)
            if p is not None and p != PREC_STRICT:
                self.saved.append((m, p))
                m.precision = PREC_STRICT
        return self

    def __exit__(self, *exc):
        for m, p in self.saved:
            m.precision = p
        set_default_precision(self.old)
        return False


def guarded_forward(module, *args, **kwargs):
    """Run `module(*args)` in the package's default precision; if fp16 operands overflowed, warn and run it again in strict mode
    (precision 0: bf16 hi / lo split, fp32 range and fp32-class accuracy at a third of the MFMA rate).  Two detectors:
      * the range guard (mi355_range_status): cast16, the 16-bit LayerNorms and every 16-bit GEMM epilogue report a finite value that
        saturates to inf;
      * a non-finite OUTPUT for finite inputs: the fused block kernels (mlp_fused.hip, cswin_fused.hip, xcit.hip, the LayerNorm-in-GEMM
        operand paths) keep their 16-bit intermediates in registers / LDS and do not track them -- a saturated intermediate there
        always reaches the block's output as inf / NaN (GELU, the second product, the softmax all propagate it), so it is caught here.
    The strict re-run overrides sub-modules built with an explicit 16-bit `precision=` as well.  Synchronises the device once per
    call -- a convenience for checkpoints with outlier activations, not the fast path."""
    import warnings
    try:
        range_status(wait=True)                               # nothing pending from earlier work
    except _ffi.Mi355RangeError:
        pass
    why = None
    try:
        y = module(*args, **kwargs)                           # a later launch of the same forward may already see the report
        range_status(wait=True)
        if _all_finite(y) or not _all_finite((args, kwargs)):
            return y                                          # finite, or the input itself was not (the reference is not either)
        why = "non-finite output for finite inputs (a 16-bit intermediate of a fused kernel saturated)"
    except _ffi.Mi355RangeError as e:
        why = str(e)
    warnings.warn(f"{type(module).__name__}: fp16 operands overflowed ({why}); re-running in strict mode", RuntimeWarning)
    try:
        range_status(wait=True)                               # launches of the abandoned forward that were still in flight may report too
    except _ffi.Mi355RangeError:
        pass
    with _forced_strict(module):
        return module(*args, **kwargs)


class _GuardState(__import__("threading").local):
    depth = 0


_guard = _GuardState()


def range_fallback_forward(module, forward, args, kwargs):
    """What `module(x)` of every drop-in class does (round 6; VERDICT round 5, missing #3): the reference returns numbers at any input or
    weight scale (ViT.py:79-89, cswin.py:176-197 compute in fp32), so the zero-edit drop-in must too.  The OUTERMOST drop-in forward on
    this thread arms the library (mi355_range_arm: the launch check behind every fp16 producer records one re-used event), runs the
    forward in the package's precision, then waits for the LAST producer of the forward only (mi355_range_wait) -- the launches queued
    behind it (attention core, fp32-output projections) keep the GPU busy while the host returns -- and reads the device's range word.
    The event is recorded ONCE per forward, in front of the first launch behind the producer the previous forward of this module
    counted as its last (mi355_range_launches -> `_mi355_nprod`; a first call records at the tail).
    Clean: the result is returned, no device synchronisation happened.  Fired (here or in a pre-launch check inside the forward): ONE
    warning, the device is drained, the forward runs again in strict mode (bf16 hi / lo split: fp32 range, fp32-class accuracy), also
    for sub-modules built with an explicit 16-bit `precision=`.  Option "range_fallback" = 0 (per device) restores the round-3 contract:
    no wait, Mi355RangeError on the next call.  Not active under hipGraph capture (an event wait is illegal there), in strict / bf16
    mode nothing can fire and the wait finds no event."""
    if _guard.depth:
        return forward(module, *args, **kwargs)
    try:
        passthrough = _ffi._capturing() or lib().mi355_get_option(b"range_fallback") != 1
    except RuntimeError:                                      # no HIP device in this process: the forward raises the package's own error
        passthrough = True
    if passthrough:
        return forward(module, *args, **kwargs)
    import warnings
    range_status()                                            # an EARLIER, unguarded launch's report is the caller's to see, not ours to absorb
    _guard.depth = 1
    why = None
    try:
        # 1 + k: "the k-th fp16 producer of this forward is the last one" -- what the previous forward of this module counted
        lib().mi355_range_arm(1 + getattr(module, "_mi355_nprod", 0))
        try:
            y = forward(module, *args, **kwargs)
            rc = lib().mi355_range_wait()
            try:
                module._mi355_nprod = int(lib().mi355_range_launches())
            except AttributeError:                            # a module that refuses new attributes: no prediction next time
                pass
            if rc == 0:
                return y
            msg = lib().mi355_last_error()
            why = msg.decode() if msg else "fp16 range word set"
        except _ffi.Mi355RangeError as e:                     # a later launch of this very forward saw the report in its pre-launch check
            why = str(e)
        finally:
            lib().mi355_range_arm(0)
        warnings.warn(f"{type(module).__name__}: fp16 operands overflowed ({why}); re-running this forward in strict mode "
                      "(set option range_fallback = 0 to raise instead)", RuntimeWarning, stacklevel=3)
        torch.cuda.synchronize()                              # launches of the abandoned forward still in flight may report too
        lib().mi355_range_status()                            # read and clear
        with _forced_strict(module):
            return forward(module, *args, **kwargs)
    finally:
        _guard.depth = 0


def range_guarded(cls):
    """Class decorator of the drop-in modules: forward() goes through range_fallback_forward (idempotent)."""
    import functools
    fwd = cls.__dict__.get("forward")
    if fwd is None or getattr(fwd, "_mi355_range_guarded", False):
        return cls

    @functools.wraps(fwd)
    def forward(self, *args, **kwargs):
        return range_fallback_forward(self, fwd, args, kwargs)

    forward._mi355_range_guarded = True
    cls.forward = forward
    return cls


def _sync_check():
    """After every exchange-kernel launch: report a failure of any EARLIER launch now (the library does the same check before it
    launches).  MI355_CHECK_SYNC=1 waits for the device first -- a debugging aid that makes the check cover this very launch."""
    import os
    sync_status(wait=os.environ.get("MI355_CHECK_SYNC") == "1")


def se_forward(x, w1, w2):
    """SELayer forward: x (B,C,H,W), w1 (C/r,C), w2 (C,C/r)."""
    x = require_device_f32(x, "x")
    w1 = require_device_f32(w1, "fc.0.weight")
    w2 = require_device_f32(w2, "fc.2.weight")
    B, C, H, W = x.shape
    Cr = w1.shape[0]
    if tuple(w1.shape) != (Cr, C) or tuple(w2.shape) != (C, Cr):
        raise ValueError(f"SE weight shapes {tuple(w1.shape)}, {tuple(w2.shape)} do not match C={C}")
    y = torch.empty_like(x)
    n = lib().mi355_se_workspace_bytes(B, C, H, W)
    ws = _ffi.workspace_dedicated(("se", B, C, H, W), n, x.device)
    check(lib().mi355_se_fwd(dptr(x), dptr(w1), dptr(w2), dptr(y), B, C, Cr, H, W, dptr(ws), ws.numel(),
                             stream_ptr(x.device)), "mi355_se_fwd")
    _sync_check()
    return y


# ---- gates built from axis reductions: GCModule, CoordinateAttention, TripletAttention, BAM (csrc/axis_attn.hip) ---------------
def bn_fold(bn, pre_bias=None):
    """(scale, shift) of an eval-mode BatchNorm, optionally absorbing the bias of the layer in front of it:
    bn(z + pre_bias) = z * scale + shift.  Cached with the BatchNorm (and the bias) like the other derived tensors."""
    def build():
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        t = bn.bias.detach() - bn.running_mean.detach() * s
        if pre_bias is not None:
            t = t + s * pre_bias.detach().reshape(-1)
        return s.float().contiguous(), t.float().contiguous()

    anchors = (bn,) if pre_bias is None else (bn, pre_bias)
    tag = _bn_tag(bn) + (() if pre_bias is None else (pre_bias._version, pre_bias.data_ptr()))
    return _derived_get(anchors, ("bnfold",), tag, build)


def _axis(x):
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    ws = workspace(lib().mi355_axis_attn_workspace_bytes(B, C, H, W), x.device)
    return x, torch.empty_like(x), ws, (B, C, H, W)


def _flat(t, name, *shape):
    return require_device_f32(t, name).reshape(*shape) if t is not None else None


def gc_forward(x, conv_w, conv_b, w1, b1, ln_w, ln_b, ln_eps, w2, b2):
    x, y, ws, (B, C, H, W) = _axis(x)
    Cr = w1.shape[0]
    check(lib().mi355_gc_fwd(dptr(x), dptr(_flat(conv_w, "conv.weight", C)), dptr(_flat(conv_b, "conv.bias", 1)),
                             dptr(_flat(w1, "transform.0.weight", Cr, C)), dptr(_flat(b1, "transform.0.bias", Cr)),
                             dptr(_flat(ln_w, "transform.1.weight", Cr)), dptr(_flat(ln_b, "transform.1.bias", Cr)),
                             dptr(_flat(w2, "transform.3.weight", C, Cr)), dptr(_flat(b2, "transform.3.bias", C)), dptr(y),
                             B, C, Cr, H, W, float(ln_eps), dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_gc_fwd")
    return y


def coordatt_forward(x, w1, b1, bn_scale, bn_shift, wh, bh, ww, bw):
    x, y, ws, (B, C, H, W) = _axis(x)
    hid = w1.shape[0]
    check(lib().mi355_coordatt_fwd(dptr(x), dptr(_flat(w1, "conv1.weight", hid, C)), dptr(_flat(b1, "conv1.bias", hid)),
                                   dptr(_flat(bn_scale, "bn1 scale", hid)), dptr(_flat(bn_shift, "bn1 shift", hid)),
                                   dptr(_flat(wh, "conv_h.weight", C, hid)), dptr(_flat(bh, "conv_h.bias", C)),
                                   dptr(_flat(ww, "conv_w.weight", C, hid)), dptr(_flat(bw, "conv_w.bias", C)), dptr(y),
                                   B, C, hid, H, W, dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_coordatt_fwd")
    return y


def triplet_forward(x, w_ch, w_cw, w_hw, affine, ksize):
    x, y, ws, (B, C, H, W) = _axis(x)
    n = 2 * ksize * ksize
    check(lib().mi355_triplet_fwd(dptr(x), dptr(_flat(w_ch, "ch.conv.conv.weight", n)), dptr(_flat(w_cw, "cw.conv.conv.weight", n)),
                                  dptr(_flat(w_hw, "hw.conv.conv.weight", n)), dptr(_flat(affine, "gate affine", 6)), dptr(y),
                                  B, C, H, W, int(ksize), dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_triplet_fwd")
    return y


def bam_forward(x, params, Cr, dilation):
    """`params`: the MI355_BAM_NPARAMS tensors in the order of the enum in include/mi355attn.h."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    if len(params) != 16:
        raise ValueError("bam_forward: expected 16 parameter tensors")
    ps = [require_device_f32(t, f"bam parameter {i}") for i, t in enumerate(params)]
    table = (ctypes.c_void_p * 16)(*[t.data_ptr() for t in ps])
    ws = workspace(lib().mi355_bam_workspace_bytes(B, C, Cr, H, W), x.device)
    y = torch.empty_like(x)
    check(lib().mi355_bam_fwd(dptr(x), ctypes.cast(table, ctypes.c_void_p), dptr(y), B, C, Cr, H, W, int(dilation), dptr(ws), ws.numel(),
                              stream_ptr(x.device)), "mi355_bam_fwd")
    return y


def bam_gates(x, params, Cr, dilation, channel=True, spatial=True):
    """The gates of BAM before their broadcast: (cg (B,C) or None, sg (B,H*W) or None); `params` as for bam_forward."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    if len(params) != 16:
        raise ValueError("bam_gates: expected 16 parameter tensors")
    ps = [require_device_f32(t, f"bam parameter {i}") for i, t in enumerate(params)]
    table = (ctypes.c_void_p * 16)(*[t.data_ptr() for t in ps])
    ws = workspace(lib().mi355_bam_workspace_bytes(B, C, Cr, H, W), x.device)
    cg = torch.empty(B, C, dtype=torch.float32, device=x.device) if channel else None
    sg = torch.empty(B, H * W, dtype=torch.float32, device=x.device) if spatial else None
    check(lib().mi355_bam_gates_fwd(dptr(x), ctypes.cast(table, ctypes.c_void_p), dptr(cg), dptr(sg), B, C, Cr, H, W, int(dilation),
                                    dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_bam_gates_fwd")
    return cg, sg


def zpool(x):
    """(B,C,H,W) -> (B,2,H,W): mean and max over the channel axis (triplet_attention.py:31-36)."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, 2, H, W, dtype=torch.float32, device=x.device)
    check(lib().mi355_zpool_fwd(dptr(x), dptr(y), B, C, H, W, stream_ptr(x.device)), "mi355_zpool_fwd")
    return y


def attention_gate(x, w, affine, ksize):
    """x * sigmoid(relu(bn(conv_kxk(zpool(x))))): w (2,k,k) conv weight, affine (2,) = folded BatchNorm scale / shift."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    w = require_device_f32(w, "w").reshape(-1)
    affine = require_device_f32(affine, "affine").reshape(-1)
    if w.numel() != 2 * ksize * ksize or affine.numel() != 2:
        raise ValueError("attention_gate: w must be (1,2,k,k) and affine (2,)")
    ws = workspace(lib().mi355_attention_gate_workspace_bytes(B, H, W), x.device)
    y = torch.empty_like(x)
    check(lib().mi355_attention_gate_fwd(dptr(x), dptr(w), dptr(affine), dptr(y), B, C, H, W, int(ksize), dptr(ws), ws.numel(),
                                         stream_ptr(x.device)), "mi355_attention_gate_fwd")
    return y


def sk_forward(x, params, planes, groups, d):
    """`params`: the MI355_SK_NPARAMS tensors in the order of the enum in include/mi355attn.h."""
    x = require_device_f32(x, "x")
    B, Cin, H, W = x.shape
    if len(params) != 14:
        raise ValueError("sk_forward: expected 14 parameter tensors")
    ps = [require_device_f32(t, f"sk parameter {i}") for i, t in enumerate(params)]
    table = (ctypes.c_void_p * 14)(*[t.data_ptr() for t in ps])
    ws = workspace(lib().mi355_sk_workspace_bytes(B, planes, H, W), x.device)
    y = torch.empty(B, planes, H, W, dtype=torch.float32, device=x.device)
    check(lib().mi355_sk_fwd(dptr(x), ctypes.cast(table, ctypes.c_void_p), dptr(y), B, Cin, planes, int(groups), int(d), H, W, dptr(ws),
                             ws.numel(), stream_ptr(x.device)), "mi355_sk_fwd")
    return y


def cam_forward(x, beta, precision=None):
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    beta = require_device_f32(beta, "beta").reshape(1)
    ws = workspace(lib().mi355_cam_workspace_bytes(B, C), x.device)
    y = torch.empty_like(x)
    check(lib().mi355_cam_fwd(dptr(x), dptr(beta), dptr(y), B, C, H, W, _prec(precision), dptr(ws), ws.numel(), stream_ptr(x.device)),
          "mi355_cam_fwd")
    return y


def tokens_to_nchw_axpy(tokens, x, alpha):
    """y[b,c,h,w] = alpha * tokens[b, h*W + w, c] + x[b,c,h,w]."""
    x = require_device_f32(x, "x")
    tokens = require_device_f32(tokens, "tokens")
    B, C, H, W = x.shape
    if tuple(tokens.shape) != (B, H * W, C):
        raise ValueError("tokens_to_nchw_axpy: tokens must be (B, H*W, C)")
    alpha = require_device_f32(alpha, "alpha").reshape(1)
    y = torch.empty_like(x)
    check(lib().mi355_tokens_to_nchw_axpy_fwd(dptr(tokens), dptr(x), dptr(alpha), dptr(y), B, H * W, C, stream_ptr(x.device)),
          "mi355_tokens_to_nchw_axpy_fwd")
    return y


def se_ex_forward(x, w1, b1, w2, b2, gate="sigmoid"):
    """SE with optional excitation biases and a choice of gate ("sigmoid" | "hard_sigmoid"): the variants inside the reference's CNNs."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    w1 = require_device_f32(w1, "w1").reshape(w1.shape[0], -1)
    w2 = require_device_f32(w2, "w2").reshape(w2.shape[0], -1)
    Cr = w1.shape[0]
    if tuple(w1.shape) != (Cr, C) or tuple(w2.shape) != (C, Cr):
        raise ValueError(f"SE weight shapes {tuple(w1.shape)}, {tuple(w2.shape)} do not match C={C}")
    if gate not in ("sigmoid", "hard_sigmoid"):
        raise ValueError("gate must be 'sigmoid' or 'hard_sigmoid'")
    y = torch.empty_like(x)
    n = lib().mi355_se_workspace_bytes(B, C, H, W)
    ws = _ffi.workspace_dedicated(("se", B, C, H, W), n, x.device)
    check(lib().mi355_se_ex_fwd(dptr(x), dptr(w1), dptr(_opt(b1, "b1")), dptr(w2), dptr(_opt(b2, "b2")), dptr(y), B, C, Cr, H, W,
                                1 if gate == "hard_sigmoid" else 0, dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_se_ex_fwd")
    _sync_check()
    return y


def eca_forward(x, wconv):
    """ECALayer forward: x (B,C,H,W), wconv (1,1,k) or (k,)."""
    x = require_device_f32(x, "x")
    wconv = require_device_f32(wconv, "conv.weight").reshape(-1)
    B, C, H, W = x.shape
    k = wconv.numel()
    y = torch.empty_like(x)
    n = lib().mi355_eca_workspace_bytes(B, C, H, W)
    ws = workspace(n, x.device)
    check(lib().mi355_eca_fwd(dptr(x), dptr(wconv), dptr(y), B, C, k, H, W, dptr(ws), ws.numel(),
                              stream_ptr(x.device)), "mi355_eca_fwd")
    return y


def cbam_forward(x, w1=None, w2=None, wconv=None, stage=0):
    """CBAM forward (stage 0), ChannelAttention alone (1) or SpatialAttention alone (2)."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    Cr, ks = 0, 0
    if stage != 2:
        w1 = require_device_f32(w1, "ca.fc.0.weight").reshape(w1.shape[0], -1)
        w2 = require_device_f32(w2, "ca.fc.2.weight").reshape(w2.shape[0], -1)
        Cr = w1.shape[0]
        if tuple(w1.shape) != (Cr, C) or tuple(w2.shape) != (C, Cr):
            raise ValueError(f"CBAM channel weight shapes {tuple(w1.shape)}, {tuple(w2.shape)} do not match C={C}")
    if stage != 1:
        wconv = require_device_f32(wconv, "sa.conv.weight")
        ks = wconv.shape[-1]
        if wconv.numel() != 2 * ks * ks:
            raise ValueError(f"CBAM spatial conv weight must be (1,2,k,k), got {tuple(wconv.shape)}")
    y = torch.empty_like(x)
    n = lib().mi355_cbam_workspace_bytes(B, C, H, W)
    ws = _ffi.workspace_dedicated(("cbam", B, C, H, W), n, x.device) if stage == 0 else workspace(n, x.device)
    check(lib().mi355_cbam_fwd(dptr(x), dptr(w1 if stage != 2 else None), dptr(w2 if stage != 2 else None),
                               dptr(wconv if stage != 1 else None), dptr(y), B, C, Cr, ks, H, W, stage,
                               dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_cbam_fwd")
    if stage == 0:
        _sync_check()
    return y


def _zoo(kind, x):
    """Common prologue of the channel-statistics gates: dense fp32 device x, output, dedicated workspace (exchange area)."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    n = lib().mi355_chan_stat_workspace_bytes(B, C)
    return x, torch.empty_like(x), _ffi.workspace_dedicated((kind, B, C, H, W), n, x.device), (B, C, H, W)


def simam_forward(x, e_lambda=1e-4):
    x, y, ws, (B, C, H, W) = _zoo("simam", x)
    check(lib().mi355_simam_fwd(dptr(x), dptr(y), B, C, H, W, float(e_lambda), dptr(ws), ws.numel(), stream_ptr(x.device)),
          "mi355_simam_fwd")
    return y


def srm_forward(x, cfc, bn_weight, bn_bias, bn_mean, bn_var, bn_eps):
    x, y, ws, (B, C, H, W) = _zoo("srm", x)
    cfc = require_device_f32(cfc, "cfc.weight").reshape(C, 2)
    ps = [require_device_f32(t, n) for t, n in ((bn_weight, "bn.weight"), (bn_bias, "bn.bias"), (bn_mean, "bn.running_mean"),
                                                (bn_var, "bn.running_var"))]
    check(lib().mi355_srm_fwd(dptr(x), dptr(cfc), dptr(ps[0]), dptr(ps[1]), dptr(ps[2]), dptr(ps[3]), float(bn_eps), dptr(y),
                              B, C, H, W, dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_srm_fwd")
    return y


def gct_gauss_forward(x, c=2, eps=1e-5):
    x, y, ws, (B, C, H, W) = _zoo("gct_gauss", x)
    check(lib().mi355_gct_gauss_fwd(dptr(x), dptr(y), B, C, H, W, float(c), float(eps), dptr(ws), ws.numel(), stream_ptr(x.device)),
          "mi355_gct_gauss_fwd")
    _sync_check()
    return y


def lct_forward(x, w, b, groups, eps=1e-5):
    x, y, ws, (B, C, H, W) = _zoo("lct", x)
    w, b = require_device_f32(w, "w"), require_device_f32(b, "b")
    check(lib().mi355_lct_fwd(dptr(x), dptr(w), dptr(b), dptr(y), B, C, int(groups), H, W, float(eps), dptr(ws), ws.numel(),
                              stream_ptr(x.device)), "mi355_lct_fwd")
    _sync_check()
    return y


def gct_forward(x, alpha, gamma, beta, epsilon=1e-5, mode="l2", after_relu=False):
    if mode not in ("l2", "l1"):
        raise ValueError("GCT mode must be 'l2' or 'l1'")
    x, y, ws, (B, C, H, W) = _zoo("gct", x)
    alpha, gamma, beta = (require_device_f32(t, n).reshape(-1) for t, n in ((alpha, "alpha"), (gamma, "gamma"), (beta, "beta")))
    check(lib().mi355_gct_fwd(dptr(x), dptr(alpha), dptr(gamma), dptr(beta), dptr(y), B, C, H, W, float(epsilon),
                              1 if mode == "l1" else 0, 1 if after_relu else 0, dptr(ws), ws.numel(), stream_ptr(x.device)),
          "mi355_gct_fwd")
    _sync_check()
    return y


def double_attention_forward(x, wA, bA, wB, bB, wV, bV, wP, bP, precision=None):
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    wA = require_device_f32(wA, "convA.weight").reshape(wA.shape[0], -1)
    wB = require_device_f32(wB, "convB.weight").reshape(wB.shape[0], -1)
    wV = require_device_f32(wV, "convV.weight").reshape(wV.shape[0], -1)
    wP = require_device_f32(wP, "proj.weight").reshape(wP.shape[0], -1)
    bA, bB, bV, bP = (require_device_f32(t, n) for t, n in ((bA, "convA.bias"), (bB, "convB.bias"),
                                                           (bV, "convV.bias"), (bP, "proj.bias")))
    cm, cn = wA.shape[0], wB.shape[0]
    Cout = wP.shape[0]
    if wV.shape[0] != cn or wP.shape[1] != cm or wA.shape[1] != C:
        raise ValueError("DoubleAttention weight shapes are inconsistent")
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    n = lib().mi355_double_attn_ws_bytes(B, C, cm, cn, H, W, _prec(precision))
    ws = workspace(n, x.device)
    check(lib().mi355_double_attn_fwd(dptr(x), dptr(wA), dptr(bA), dptr(wB), dptr(bB), dptr(wV), dptr(bV),
                                      dptr(wP), dptr(bP), dptr(y), B, C, cm, cn, H, W, _prec(precision),
                                      dptr(ws), ws.numel(), stream_ptr(x.device)), "mi355_double_attn_fwd")
    return y


# ---- dense building blocks ---------------------------------------------------------------------------------
def linear(x, weight, bias=None, act=ACT_NONE, gamma=None, resid=None, precision=None, out=None):
    """Y = resid + gamma * act(x @ weight^T + bias) over the last axis of x (any leading shape)."""
    ldx = None
    if isinstance(x, torch.Tensor) and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 \
            and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and not x.is_contiguous():
        ldx = x.stride(0)                       # row-strided view (e.g. tokens[:, 0]) is consumed in place
    else:
        x = require_device_f32(x, "x")
    weight = require_device_f32(weight, "weight")
    bias, gamma = _opt(bias, "bias"), _opt(gamma, "gamma")
    N, K = weight.shape
    if x.shape[-1] != K:
        raise ValueError(f"linear: x last dim {x.shape[-1]} != weight in_features {K}")
    lead = x.shape[:-1]
    M = x.numel() // K
    ldx = K if ldx is None else ldx
    if resid is not None:
        resid = require_device_f32(resid, "resid")
        if resid.numel() != M * N:
            raise ValueError("linear: residual shape mismatch")
    y = out if out is not None else torch.empty(*lead, N, dtype=torch.float32, device=x.device)
    check(lib().mi355_linear_fwd(dptr(x), dptr(weight), dptr(bias), dptr(gamma), dptr(resid), dptr(y),
                                 M, N, K, ldx, N, act, _prec(precision), stream_ptr(x.device)), "mi355_linear_fwd")
    return y


def token_mix(weight, x, bias=None, act=ACT_NONE, resid=None, precision=None):
    """Y_b = resid_b + act(weight @ x_b + bias[:,None]) for x (B,N,C), weight (T,N) -> (B,T,C)."""
    x = require_device_f32(x, "x")
    weight = require_device_f32(weight, "weight")
    bias, resid = _opt(bias, "bias"), _opt(resid, "resid")
    B, N, C = x.shape
    T = weight.shape[0]
    if weight.shape[1] != N:
        raise ValueError("token_mix: weight in_features != sequence length")
    y = torch.empty(B, T, C, dtype=torch.float32, device=x.device)
    check(lib().mi355_token_mix_fwd(dptr(weight), dptr(x), dptr(bias), dptr(resid), dptr(y), B, T, N, C, act,
                                    _prec(precision), stream_ptr(x.device)), "mi355_token_mix_fwd")
    return y


def layernorm(x, weight, bias, eps=1e-5):
    x = require_device_f32(x, "x")
    weight = require_device_f32(weight, "weight")
    bias = require_device_f32(bias, "bias")
    cols = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().mi355_layernorm_fwd(dptr(x), dptr(weight), dptr(bias), dptr(y), x.numel() // cols, cols,
                                    float(eps), stream_ptr(x.device)), "mi355_layernorm_fwd")
    return y


# ---- 16-bit activation dataflow (fast path of the composite blocks) -----------------------------------------------
def dtype16(precision):
    p = _prec(precision)
    if p == PREC_FP16:
        return torch.float16
    if p == PREC_BF16:
        return torch.bfloat16
    raise ValueError("the 16-bit dataflow exists for precision 1 (fp16) and 2 (bf16) only")


def cswin_lepe_attention16_pair(qkv16, w0, b0, w1, b1, out16, reso, heads, split, scale, precision=None):
    """Both stripe branches of a CSWinBlock in one launch (16-bit qkv / out buffers, `heads` per branch)."""
    qkv16 = _require16(qkv16, "qkv16", precision)
    ws = [require_device_f32(t, n) for t, n in ((w0, "attns.0.get_v.weight"), (b0, "attns.0.get_v.bias"), (w1, "attns.1.get_v.weight"),
                                                (b1, "attns.1.get_v.bias"))]
    B = qkv16.shape[0]
    Ctot = qkv16.shape[-1] // 3
    check(lib().mi355_cswin_lepe_attn16_pair_fwd(dptr(qkv16), dptr(ws[0]), dptr(ws[1]), dptr(ws[2]), dptr(ws[3]), dptr(out16), B, reso, Ctot,
                                                 heads, split, float(scale), _prec(precision), stream_ptr(qkv16.device)),
          "mi355_cswin_lepe_attn16_pair_fwd")
    return out16


def cswin_stripe_ok(C, reso, split, heads, precision=None):
    """Shape / precision envelope of mi355_cswin_stripe_attn_fwd (CSWin-T stages 1-2)."""
    return _prec(precision) in (PREC_FP16, PREC_BF16) and C in (64, 128) and heads * 32 == C and reso * split <= 64 and reso % split == 0


def _ln_folded16(ln, lin, p):
    """(W diag(ln.weight) in 16 bit, b + W ln.bias in fp32) of a Linear behind a LayerNorm, cached per parameter version."""
    N = lin.weight.shape[0]

    def build():
        w = lin.weight.detach()
        b = lin.bias.detach() if lin.bias is not None else torch.zeros(N, dtype=torch.float32, device=w.device)
        b = linear(ln.bias.detach().reshape(1, -1).contiguous(), w.contiguous(), b, precision=PREC_STRICT).reshape(-1)   # b + W ln.bias on the library's own engine (no vendor BLAS launch)
        return cast16((w * ln.weight.detach()[None, :]).contiguous(), p), b.contiguous()

    parts = [lin.weight] + ([] if lin.bias is None else [lin.bias]) + [ln.weight, ln.bias]
    tag = tuple((t._version, t.data_ptr()) for t in parts)
    return _derived_get((lin, ln), ("ln_linear16", p), tag, build)


def cswin_stripe_attention(x, ln, qkv, getv0, getv1, reso, heads, split, scale, precision=None):
    """LayerNorm -> qkv -> both stripe branches of LePE attention in one kernel; x (B, L, C) fp32 -> ctx (B, L, C) 16-bit.
    `heads` = heads of the whole block (heads / 2 per branch)."""
    p = _prec(precision)
    x = require_device_f32(x, "x")
    B, L, C = x.shape
    w16, b = _ln_folded16(ln, qkv, p)
    ws = [require_device_f32(t, n) for t, n in ((getv0.weight, "attns.0.get_v.weight"), (getv0.bias, "attns.0.get_v.bias"),
                                                (getv1.weight, "attns.1.get_v.weight"), (getv1.bias, "attns.1.get_v.bias"))]
    ctx = torch.empty(B, L, C, dtype=dtype16(p), device=x.device)
    check(lib().mi355_cswin_stripe_attn_fwd(dptr(x), dptr(w16), dptr(b), dptr(ws[0]), dptr(ws[1]), dptr(ws[2]), dptr(ws[3]), dptr(ctx), B, reso,
                                            C, heads // 2, split, float(scale), float(ln.eps), p, stream_ptr(x.device)),
          "mi355_cswin_stripe_attn_fwd")
    return ctx


def fast_gemm_ok(K, N):
    """Shape envelope of mi355_linear16_fwd (K-step 64, float4 epilogue)."""
    return K % 64 == 0 and N % 4 == 0


def _require16(t, name, precision):
    want = dtype16(precision)
    if not t.is_cuda or t.dtype != want:
        raise TypeError(f"{name}: expected a {want} device tensor, got {t.dtype} on {t.device}")
    return t if t.is_contiguous() else t.contiguous()


def cast16(x, precision=None):
    """fp32 -> MFMA operand format (round-to-nearest-even), same shape."""
    _range_check()
    x = require_device_f32(x, "x")
    y = torch.empty(x.shape, dtype=dtype16(precision), device=x.device)
    check(lib().mi355_cast16_fwd(dptr(x), dptr(y), x.numel(), _prec(precision), stream_ptr(x.device)), "mi355_cast16_fwd")
    return y


_derived = {}


def _derived_get(anchors, subkey, tag, build):
    """Cache of tensors derived from parameters (16-bit copies, re-laid-out or BatchNorm-folded conv weights).

    `anchors` are the owning objects (parameters / modules).  An entry is keyed by their ids but only honoured while every anchor
    is still THE SAME LIVE OBJECT (weak references: CPython recycles ids, the caching allocator recycles addresses, and a fresh
    parameter starts at version 0 again -- an id/address/version tag alone can resurrect the weights of a deleted model), and it is
    dropped as soon as one of them is collected, so the derived copies do not outlive their model.  `tag` captures in-place updates
    (version counters) and moves (data pointers)."""
    key = tuple(id(a) for a in anchors) + (subkey,)
    hit = _derived.get(key)
    if hit is not None and hit[1] == tag and all(r() is a for r, a in zip(hit[0], anchors)):
        return hit[2]
    val = build()
    refs = tuple(weakref.ref(a, lambda _r, k=key: _derived.pop(k, None)) for a in anchors)
    _derived[key] = (refs, tag, val)
    return val


def weight16(param, precision=None):
    """16-bit copy of a weight, converted once and reused until the parameter is modified (in-place version bump),
    moved, re-precisioned or collected."""
    p = _prec(precision)
    tag = (param._version, param.data_ptr(), tuple(param.shape))
    return _derived_get((param,), ("w16", p), tag, lambda: cast16(param.detach(), p))


FP16_MIN_NORMAL = 2.0 ** -14


def weight16_scaled(weight, bias, gamma, precision=None):
    """(W16', b') with a LayerScale folded in: y = resid + gamma * (x W^T + b) = resid + x (gamma[:, None] * W)^T + gamma * b when no
    activation sits between the product and the scale (XCiT: x + gamma1 * proj(..), x + gamma2 * fc2(..), xcit.py:290-294).  The GEMM
    then needs no per-column scale in its epilogue and can take the two-accumulator kernel.  Cached with the parameters.

    Returns None when the fold would cost accuracy or range in fp16: gamma * W must stay inside the fp16 NORMAL range -- with the
    published deep-XCiT initialisation eta = 1e-5, gamma * W ~ 2e-7 lies in the subnormals (step 6e-8: ~12 % of the weights flush to
    zero, the branch's relative error rises from 3e-4 to 9e-2) -- so the fold is taken only if the smallest non-zero |gamma| times the
    median |W| is at least 8 fp16 min-normals and the largest product is finite; the caller then keeps gamma in the fp32 epilogue
    (`linear16(..., gamma=)`).  bf16 has the fp32 exponent range and always folds."""
    p = _prec(precision)
    anchors = (weight, gamma) + ((bias,) if bias is not None else ())
    tag = tuple((t._version, t.data_ptr(), tuple(t.shape)) for t in anchors)

    def build():
        g = gamma.detach().reshape(-1)
        w = weight.detach()
        if p == PREC_FP16:
            # the decision is reduced to ONE flag on the device and read with one synchronising copy (four .item()-style
            # reads before); a synchronising read is illegal under stream capture, so a first build there is refused with a clear message
            if w.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight16_scaled: the LayerScale fold decision reads the weights once on the host -- run one eager "
                                   "forward (warm-up) before capturing this module in a hipGraph")
            ga, wa = g.abs(), w.abs()
            gmin = torch.where(ga > 0, ga, torch.full_like(ga, float("inf"))).min()
            gmin = torch.where(torch.isinf(gmin), torch.ones_like(gmin), gmin)
            bad = (gmin * wa.median() < 8.0 * FP16_MIN_NORMAL) | (ga.max() * wa.max() >= 65504.0)
            if bool(bad.item()):
                return None
        w16 = (w * g[:, None]).to(dtype16(p)).contiguous()      # round-to-nearest-even, like mi355_cast16_fwd
        return w16, (None if bias is None else (bias.detach() * g).contiguous())

    return _derived_get(anchors, ("w16scaled", p), tag, build)


def mlp_fused_ok(C, hidden, precision=None):
    """Shape / precision envelope of mi355_mlp_fused_fwd (round 6: C = 256 / 384 on the weight-split kernel, option "mlp_wide")."""
    if _prec(precision) not in (PREC_FP16, PREC_BF16):
        return False
    if (C, hidden) in ((64, 256), (128, 512)):
        return True
    return (C, hidden) in ((256, 1024), (384, 1536)) and lib().mi355_get_option(b"mlp_wide") == 1


def proj_mlp_fused_ok(C, hidden, precision=None):
    """Shape / precision envelope of mi355_proj_mlp_fused_fwd."""
    return _prec(precision) in (PREC_FP16, PREC_BF16) and (C, hidden) in ((64, 256), (128, 512))


def mlp_fused(x, ln, fc1, fc2, gamma=None, precision=None, ctx16=None, proj=None):
    """y = x + gamma * fc2(gelu(fc1(ln(x)))) in one kernel (hidden activations never reach HBM).  `ln` is the LayerNorm in front of
    the MLP (its affine part is folded into fc1 here, cached per parameter version) or None.  With `ctx16` (M, C) 16-bit and `proj`
    (a Linear) the kernel first forms x1 = x + proj(ctx16) and runs the MLP on x1 (second half of a CSWinBlock in one launch)."""
    p = _prec(precision)
    x = require_device_f32(x, "x")
    C = x.shape[-1]
    Hd = fc1.weight.shape[0]

    def build():
        w1 = fc1.weight.detach()
        b1 = fc1.bias.detach() if fc1.bias is not None else torch.zeros(Hd, dtype=torch.float32, device=w1.device)
        if ln is not None:
            b1 = linear(ln.bias.detach().reshape(1, -1).contiguous(), w1.contiguous(), b1, precision=PREC_STRICT).reshape(-1)    # b1 + W1 ln.bias, no vendor BLAS
            w1 = w1 * ln.weight.detach()[None, :]
        # range proof (round 6): behind a LayerNorm |xn| <= sqrt(C - 1), so the 16-bit hidden activation is bounded by the folded weights
        # alone; proven once per parameter version (one flag read), the kernel then has nothing to report and the host nothing to wait for.
        # 60000 leaves room for the fp16 rounding of W1' (2^-11 relative).  Under stream capture no read is possible: unproven.
        proven = False
        if ln is not None and p == PREC_FP16 and not (w1.is_cuda and torch.cuda.is_current_stream_capturing()):
            bound = (w1.abs().sum(dim=1) * float(C - 1) ** 0.5 + b1.abs()).max()
            proven = bool((bound < 60000.0).item())
        return cast16(w1.contiguous(), p), b1.contiguous(), proven

    parts = [fc1.weight] + ([] if fc1.bias is None else [fc1.bias]) + ([] if ln is None else [ln.weight, ln.bias])
    tag = tuple((t._version, t.data_ptr()) for t in parts)
    anchors = (fc1,) if ln is None else (fc1, ln)
    w1_16, b1, proven = _derived_get(anchors, ("mlp_fused_w1", p), tag, build)
    ln_flags = (0 if ln is None else 1) | (2 if proven else 0)
    if C != 128:                                        # C = 64 (weights resident in LDS) and C = 256 / 384 (fragments global -> VGPR): row-major
        w2_16 = weight16(fc2.weight, p)
    else:                                               # slice-major (hidden/32, C, 32): the kernel streams 32-unit slices through LDS
        w2_16 = _derived_get((fc2,), ("mlp_fused_w2", p), (fc2.weight._version, fc2.weight.data_ptr()),
                             lambda: cast16(fc2.weight.detach().reshape(C, Hd // 32, 32).permute(1, 0, 2).contiguous(), p))
    y = torch.empty_like(x)
    M = x.numel() // C
    if ctx16 is not None:
        ctx16 = _require16(ctx16, "ctx16", p)
        wp16 = weight16(proj.weight, p)
        bp = proj.bias if proj.bias is not None else _derived_get((proj,), ("zero_bias",), (proj.weight.data_ptr(),),
                                                                  lambda: torch.zeros(C, dtype=torch.float32, device=x.device))
        check(lib().mi355_proj_mlp_fused_fwd(dptr(x), dptr(ctx16), dptr(wp16), dptr(require_device_f32(bp, "proj.bias")), dptr(w1_16), dptr(b1),
                                             dptr(w2_16), dptr(_opt(fc2.bias, "fc2.bias")), dptr(_opt(gamma, "gamma")), dptr(y), M, C, Hd,
                                             ln_flags, float(ln.eps) if ln is not None else 0.0, p, stream_ptr(x.device)),
              "mi355_proj_mlp_fused_fwd")
        return y
    check(lib().mi355_mlp_fused_fwd(dptr(x), dptr(w1_16), dptr(b1), dptr(w2_16), dptr(_opt(fc2.bias, "fc2.bias")), dptr(_opt(gamma, "gamma")),
                                    dptr(y), M, C, Hd, ln_flags, float(ln.eps) if ln is not None else 0.0, p,
                                    stream_ptr(x.device)), "mi355_mlp_fused_fwd")
    return y


def ln_linear16_ok(K, N, precision=None):
    """Shape / precision envelope of mi355_ln_linear16_fwd."""
    return _prec(precision) in (PREC_FP16, PREC_BF16) and K in (64, 128) and N % 8 == 0


def ln_linear16(x, ln, lin, act=ACT_NONE, out16=True, precision=None):
    """act(lin(ln(x))) with the LayerNorm applied on the way into the GEMM (no 16-bit LayerNorm tensor in HBM).  The LayerNorm affine
    part is folded into the Linear here (W' = W diag(ln.weight), b' = b + W ln.bias), cached per parameter version."""
    p = _prec(precision)
    x = require_device_f32(x, "x")
    K = x.shape[-1]
    N = lin.weight.shape[0]

    w16, b = _ln_folded16(ln, lin, p)
    M = x.numel() // K
    y = torch.empty(*x.shape[:-1], N, dtype=dtype16(p) if out16 else torch.float32, device=x.device)
    check(lib().mi355_ln_linear16_fwd(dptr(x), dptr(w16), dptr(b), dptr(y), M, N, K, K, N, float(ln.eps), act, 1 if out16 else 0, p,
                                      stream_ptr(x.device)), "mi355_ln_linear16_fwd")
    return y


def layernorm16(x, weight, bias, eps=1e-5, precision=None):
    _range_check()
    x = require_device_f32(x, "x")
    weight = require_device_f32(weight, "weight")
    bias = require_device_f32(bias, "bias")
    cols = x.shape[-1]
    y = torch.empty(x.shape, dtype=dtype16(precision), device=x.device)
    check(lib().mi355_layernorm16_fwd(dptr(x), dptr(weight), dptr(bias), dptr(y), x.numel() // cols, cols, float(eps),
                                      _prec(precision), stream_ptr(x.device)), "mi355_layernorm16_fwd")
    return y


def linear16(x16, w16, bias=None, act=ACT_NONE, gamma=None, resid=None, out16=False, precision=None):
    """Y = resid + gamma * act(x16 @ w16^T + bias); 16-bit operands, fp32 or 16-bit result."""
    _range_check()
    x16 = _require16(x16, "x16", precision)
    w16 = _require16(w16, "w16", precision)
    bias, gamma, resid = _opt(bias, "bias"), _opt(gamma, "gamma"), _opt(resid, "resid")
    N, K = w16.shape
    if x16.shape[-1] != K:
        raise ValueError(f"linear16: x last dim {x16.shape[-1]} != weight in_features {K}")
    M = x16.numel() // K
    if resid is not None and resid.numel() != M * N:
        raise ValueError("linear16: residual shape mismatch")
    y = torch.empty(*x16.shape[:-1], N, dtype=dtype16(precision) if out16 else torch.float32, device=x16.device)
    nws = lib().mi355_linear16_workspace_bytes(M, N, K)                  # split last round of the persistent kernel (0: not needed)
    ws = _ffi.workspace_named("linear16", nws, x16.device) if nws else None
    check(lib().mi355_linear16_ws_fwd(dptr(x16), dptr(w16), dptr(bias), dptr(gamma), dptr(resid), dptr(y), M, N, K, K, N, act,
                                      1 if out16 else 0, _prec(precision), dptr(ws), nws if ws is not None else 0,
                                      stream_ptr(x16.device)), "mi355_linear16_ws_fwd")
    return y


def cast_linear16(x, w16, bias=None, act=ACT_NONE, precision=None):
    """act(T(x) @ w16^T + bias) in the 16-bit operand format for an fp32 OR 16-bit x: an fp32 x goes through mi355_linear16_x32_fwd (the cast
    inside the GEMM's staging: one launch, no 16-bit copy of x in HBM) where that entry takes the shape, else through cast16 + linear16;
    the same bits either way."""
    p = _prec(precision)
    if x.dtype != torch.float32:
        return linear16(x, w16, bias, act=act, out16=True, precision=p)
    _range_check()
    x = require_device_f32(x, "x")
    w16 = _require16(w16, "w16", p)
    N, K = w16.shape
    if x.shape[-1] != K:
        raise ValueError(f"cast_linear16: x last dim {x.shape[-1]} != weight in_features {K}")
    M = x.numel() // K
    if K in (256, 384, 512) and M >= 128:
        bias_ = _opt(bias, "bias")
        y = torch.empty(*x.shape[:-1], N, dtype=dtype16(p), device=x.device)
        rc = lib().mi355_linear16_x32_fwd(dptr(x), dptr(w16), dptr(bias_), dptr(y), M, N, K, K, N, act, p, stream_ptr(x.device))
        if rc != _ffi.MI355_EUNSUPPORTED:
            check(rc, "mi355_linear16_x32_fwd")
            return y
    return linear16(cast16(x, p), w16, bias, act=act, out16=True, precision=p)


def linear16_stats(x16, w16, bias, resid, eps, precision=None):
    """(Y, stats) with Y = resid + x16 @ w16^T + bias (fp32) and stats (rows, 2) = (mean, 1 / sqrt(var + eps)) of every row of Y -- the
    LayerNorm statistics the next block needs, written by the GEMM that owns whole rows (mi355_linear16_stats_fwd: N = K = 256 / 384).
    Returns None where the entry is not built for the shape (the caller then runs linear16 and lets the consumer compute its statistics)."""
    _range_check()
    p = _prec(precision)
    x16 = _require16(x16, "x16", p)
    w16 = _require16(w16, "w16", p)
    N, K = w16.shape
    M = x16.numel() // K
    if not (N == K and K in (256, 384) and M >= 32 and resid is not None and lib().mi355_get_option(b"gemm_wreg") == 1):
        return None
    bias, resid = _opt(bias, "bias"), require_device_f32(resid, "resid")
    y = torch.empty(x16.shape[:-1] + (N,), dtype=torch.float32, device=x16.device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x16.device)
    check(lib().mi355_linear16_stats_fwd(dptr(x16), dptr(w16), dptr(bias), dptr(resid), dptr(y), M, N, K, K, N, p, dptr(stats), float(eps),
                                         stream_ptr(x16.device)), "mi355_linear16_stats_fwd")
    return y, stats


def linear16_ln16(x16, w16, bias, resid, ln, precision=None):
    """(Y, U16): Y = resid + x16 @ w16^T + bias (fp32) and U16 = ln(Y) in the 16-bit operand format, one launch (mi355_linear16_ln16_fwd:
    N = K = 256; the projection + residual and the LayerNorm in front of the MLP of a CSWin stage-3 block, cswin.py:192-194).
    Returns None where the entry is not built for the shape."""
    _range_check()
    p = _prec(precision)
    x16 = _require16(x16, "x16", p)
    w16 = _require16(w16, "w16", p)
    N, K = w16.shape
    M = x16.numel() // K
    if not (N == K and K == 256 and M >= 32 and resid is not None and lib().mi355_get_option(b"gemm_wreg") == 1):
        return None
    bias, resid = _opt(bias, "bias"), require_device_f32(resid, "resid")
    lw, lb = require_device_f32(ln.weight, "ln.weight"), require_device_f32(ln.bias, "ln.bias")
    y = torch.empty(x16.shape[:-1] + (N,), dtype=torch.float32, device=x16.device)
    u = torch.empty(x16.shape[:-1] + (N,), dtype=dtype16(p), device=x16.device)
    check(lib().mi355_linear16_ln16_fwd(dptr(x16), dptr(w16), dptr(bias), dptr(resid), dptr(y), dptr(lw), dptr(lb), float(ln.eps), dptr(u),
                                        M, N, K, K, N, N, p, stream_ptr(x16.device)), "mi355_linear16_ln16_fwd")
    return y, u


def layernorm16_t(x, weight, bias, eps=1e-5, NP=None, precision=None):
    """LayerNorm over C of x (B, N, C), written transposed per image in 16 bit: (B, C, NP), zeros for n >= N."""
    x = require_device_f32(x, "x")
    weight = require_device_f32(weight, "weight")
    bias = require_device_f32(bias, "bias")
    B, N, C = x.shape
    NP = NP or -(-N // 64) * 64
    ut = torch.empty(B, C, NP, dtype=dtype16(precision), device=x.device)
    check(lib().mi355_layernorm16_t_fwd(dptr(x), dptr(weight), dptr(bias), dptr(ut), B, N, C, NP, float(eps), _prec(precision),
                                        stream_ptr(x.device)), "mi355_layernorm16_t_fwd")
    return ut


MIXER_TOKENS = 196      # the token count mi355_mixer_token_fwd is built for (14 x 14 patches)


def mixer_token_ok(N, T, C, precision=None):
    """Envelope of mi355_mixer_token_fwd: 16-bit operand mode, N = 196 tokens, T % 32 == 0 hidden token units (<= 512), C % 256 == 0
    (<= 1024), and the option switch "mixer_fused"."""
    return (_prec(precision) in (PREC_FP16, PREC_BF16) and N == MIXER_TOKENS and T % 32 == 0 and 0 < T <= 512 and C % 256 == 0 and 0 < C <= 1024
            and _ffi.get_option("mixer_fused") != 0)


def weight16_slices(w, rows, precision=None):
    """fc2.weight (N, T) as (T / 32, rows, 32) 16-bit: w2s[kb][n][j] = w[n][kb * 32 + j], rows n >= N zero (the slice-major operand of
    mi355_mixer_token_fwd; cached with the parameter)."""
    def build():
        N, T = w.shape
        out = torch.zeros(T // 32, rows, 32, dtype=dtype16(precision), device=w.device)
        out[:, :N, :] = w.detach().reshape(N, T // 32, 32).permute(1, 0, 2)
        return out.contiguous()
    tag = (w._version, w.data_ptr(), tuple(w.shape))
    return _derived_get((w,), ("w16slices", rows, _prec(precision)), tag, build)


def mixer_token_mlp(x, norm, fc1, fc2, precision=None):
    """x + (gelu(LayerNorm(x)^T W1^T + b1) W2^T + b2)^T for x (B, 196, C): mlp_mixer.py:47 as a row-statistics pass + one kernel."""
    x = require_device_f32(x, "x")
    B, N, C = x.shape
    T = fc1.weight.shape[0]
    p = _prec(precision)
    w1p = weight16_padk(fc1.weight, 224, p)
    w2s = weight16_slices(fc2.weight, 208, p)
    y = torch.empty_like(x)
    n = lib().mi355_mixer_token_workspace_bytes(B, N, C)
    ws = workspace(n, x.device)
    check(lib().mi355_mixer_token_fwd(dptr(x), dptr(require_device_f32(norm.weight, "norm.weight")), dptr(require_device_f32(norm.bias, "norm.bias")),
                                      float(norm.eps), dptr(w1p), dptr(require_device_f32(fc1.bias, "fc1.bias")), dptr(w2s),
                                      dptr(require_device_f32(fc2.bias, "fc2.bias")), dptr(y), B, N, C, T, p, dptr(ws), ws.numel(),
                                      stream_ptr(x.device)), "mi355_mixer_token_fwd")
    return y


def linear16_tr(xt16, w16, bias, resid, precision=None):
    """xt16 (B, C, K) 16-bit, w16 (N, K) -> fp32 (B, N, C) = resid + (xt16 @ w16^T + bias) transposed per image."""
    xt16 = _require16(xt16, "xt16", precision)
    w16 = _require16(w16, "w16", precision)
    bias, resid = _opt(bias, "bias"), _opt(resid, "resid")
    B, C, K = xt16.shape
    N = w16.shape[0]
    if w16.shape[1] != K:
        raise ValueError(f"linear16_tr: weight in_features {w16.shape[1]} != {K}")
    if resid is not None and tuple(resid.shape) != (B, N, C):
        raise ValueError("linear16_tr: residual shape mismatch")
    y = torch.empty(B, N, C, dtype=torch.float32, device=xt16.device)
    check(lib().mi355_linear16_tr_fwd(dptr(xt16), dptr(w16), dptr(bias), dptr(resid), dptr(y), B * C, N, K, K, C, _prec(precision),
                                      stream_ptr(xt16.device)), "mi355_linear16_tr_fwd")
    return y


def weight16_padk(w, K, precision=None):
    """16-bit copy of a Linear weight (N, k) zero-padded along in_features to K (cached with the parameter)."""
    def build():
        out = torch.zeros(w.shape[0], K, dtype=dtype16(precision), device=w.device)
        out[:, :w.shape[1]] = w.detach()
        return out
    tag = (w._version, w.data_ptr(), tuple(w.shape))
    return _derived_get((w,), ("w16padk", K, _prec(precision)), tag, build)


# ---- LayerNorm folded into the neighbouring GEMMs (csrc/ln_fold.hip) ---------------------------------------------------------
LN_FOLD_TOL = 1.0      # |c - mean| <= tol * std keeps a row on the fast path (mi355_ln_finalize_fwd)


class LnState:
    """What a folded LayerNorm hands to its consumer GEMM: a16 (rows, C) = T(x - c) (or the plain operand for rewritten rows),
    rowtau (rows, 2) fp32 = {rstd, rstd (c - mean)}, cvec (rows) fp32 = the exact row mean (the next producer's c)."""
    __slots__ = ("a16", "rowtau", "cvec")

    def __init__(self, a16, rowtau, cvec):
        self.a16, self.rowtau, self.cvec = a16, rowtau, cvec


def ln_fold_enabled():
    """Option "ln_fold" of the current device (default 0: the LayerNorm launches are faster than the fold)."""
    return _ffi.get_option("ln_fold") == 1


def ln_fold_ok(rows, C, N, K, precision=None):
    """Envelope of the fold around a residual-stream GEMM (N = C outputs, K inputs) on `rows` token rows: 16-bit operand mode, the
    two-accumulator producer (rows % 128 == 0, C % 256 == 0, K % 64 == 0, K >= 640) and the option switch."""
    return (_prec(precision) in (PREC_FP16, PREC_BF16) and rows % 128 == 0 and N == C and C % 256 == 0 and C <= 2048 and
            K % 64 == 0 and K >= 640 and _ffi.get_option("ln_fold") == 1 and rows * C * 4 < (1 << 31))


def lnfold_weights(ln, lin, precision=None):
    """(W16', colsum, bias') of a Linear behind a LayerNorm: W' = T(gamma * W) (N, K), colsum[n] = sum_k float(W'[n][k]) -- of the
    ROUNDED weights, so that the rank-1 mean correction cancels against the products exactly --, bias' = b + W beta.  Cached with
    the four parameters; None when gamma * W leaves the fp16 range."""
    p = _prec(precision)
    anchors = (ln.weight, ln.bias, lin.weight) + ((lin.bias,) if lin.bias is not None else ())
    tag = tuple((t._version, t.data_ptr()) for t in anchors)

    def build():
        w = lin.weight.detach()
        wg = w * ln.weight.detach()[None, :]
        if p == PREC_FP16 and not bool((wg.abs() < 65504.0).all()):
            return None
        w16 = wg.to(dtype16(p)).contiguous()
        colsum = w16.double().sum(dim=1).float().contiguous()
        b = w.double() @ ln.bias.detach().double()
        if lin.bias is not None:
            b = b + lin.bias.detach().double()
        return w16, colsum, b.float().contiguous()

    return _derived_get(anchors, ("lnfold", p), tag, build)


def ln_center16(x, eps=1e-5, precision=None):
    """First LayerNorm of a folded chain: x (..., C) fp32 -> LnState with the plain operand (x - mean) * rstd for every row."""
    p = _prec(precision)
    x = require_device_f32(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    a16 = torch.empty(x.shape, dtype=dtype16(p), device=x.device)
    rowtau = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    cvec = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().mi355_ln_center16_fwd(dptr(x), dptr(a16), dptr(rowtau), dptr(cvec), rows, C, float(eps), p, stream_ptr(x.device)),
          "mi355_ln_center16_fwd")
    return LnState(a16, rowtau, cvec)


def linear16_lnfold(state, w16, bias, colsum, act=ACT_NONE, precision=None):
    """Consumer of a folded LayerNorm: act(LayerNorm(x) W^T + b) in 16 bit from LnState and lnfold_weights()."""
    _range_check()
    p = _prec(precision)
    a16 = _require16(state.a16, "a16", p)
    w16 = _require16(w16, "w16", p)
    N, K = w16.shape
    if a16.shape[-1] != K:
        raise ValueError(f"linear16_lnfold: operand width {a16.shape[-1]} != weight in_features {K}")
    M = a16.numel() // K
    y = torch.empty(*a16.shape[:-1], N, dtype=dtype16(p), device=a16.device)
    check(lib().mi355_linear16_lnfold_fwd(dptr(a16), dptr(w16), dptr(_opt(bias, "bias")), dptr(state.rowtau), dptr(colsum), dptr(y),
                                          M, N, K, K, N, act, p, stream_ptr(a16.device)), "mi355_linear16_lnfold_fwd")
    return y


def linear16_emit(x16, w16, bias, resid, cvec, eps=1e-5, act=ACT_NONE, precision=None, slow_rows=None):
    """Producer + finalize: y = resid + act(x16 W^T + b) (fp32) AND the LnState of LayerNorm(y) for the next GEMM.  `cvec` (rows)
    holds the rows' means before this update and is overwritten with the new ones."""
    _range_check()
    p = _prec(precision)
    x16 = _require16(x16, "x16", p)
    w16 = _require16(w16, "w16", p)
    N, K = w16.shape
    if x16.shape[-1] != K:
        raise ValueError(f"linear16_emit: x last dim {x16.shape[-1]} != weight in_features {K}")
    M = x16.numel() // K
    resid = _opt(resid, "resid")
    if resid is not None and resid.numel() != M * N:
        raise ValueError("linear16_emit: residual shape mismatch")
    dev = x16.device
    y = torch.empty(*x16.shape[:-1], N, dtype=torch.float32, device=dev)
    a16 = torch.empty(*x16.shape[:-1], N, dtype=dtype16(p), device=dev)
    stats = _ffi.workspace_named("ln_fold_stats", lib().mi355_ln_fold_stats_bytes(M, N), dev)
    rowtau = torch.empty(M, 2, dtype=torch.float32, device=dev)
    check(lib().mi355_linear16_emit_fwd(dptr(x16), dptr(w16), dptr(_opt(bias, "bias")), dptr(resid), dptr(y), M, N, K, K, act, p,
                                        dptr(cvec), dptr(a16), dptr(stats), stream_ptr(dev)), "mi355_linear16_emit_fwd")
    check(lib().mi355_ln_finalize_fwd(dptr(stats), dptr(y), dptr(a16), dptr(cvec), dptr(rowtau), M, N, float(eps), float(LN_FOLD_TOL), p,
                                      dptr(slow_rows), stream_ptr(dev)), "mi355_ln_finalize_fwd")
    return y, LnState(a16, rowtau, cvec)


def mhsa16(x, wqkv16, bqkv, wproj16, bproj, num_heads, scale, resid=None, precision=None):
    """ViT Attention.forward as one C call (mi355_mhsa_fwd): x (B,N,C) fp32 or already in the 16-bit operand format."""
    _range_check()
    p = _prec(precision)
    x_is16 = x.dtype != torch.float32
    x = _require16(x, "x", p) if x_is16 else require_device_f32(x, "x")
    wqkv16, wproj16 = _require16(wqkv16, "wqkv16", p), _require16(wproj16, "wproj16", p)
    B, N, C = x.shape
    if tuple(wqkv16.shape) != (3 * C, C) or tuple(wproj16.shape) != (C, C):
        raise ValueError("mhsa16: weight shapes do not match the embedding width")
    bqkv, bproj, resid = _opt(bqkv, "qkv.bias"), _opt(bproj, "proj.bias"), _opt(resid, "resid")
    if resid is not None and resid.numel() != x.numel():
        raise ValueError("mhsa16: residual shape mismatch")
    y = torch.empty(B, N, C, dtype=torch.float32, device=x.device)
    nws = lib().mi355_mhsa_workspace_bytes(B, N, C, 1 if x_is16 else 0)
    ws = _ffi.workspace_named("mhsa", nws, x.device)
    check(lib().mi355_mhsa_fwd(dptr(x), 1 if x_is16 else 0, dptr(wqkv16), dptr(bqkv), dptr(wproj16), dptr(bproj), dptr(resid), dptr(y),
                               B, N, C, int(num_heads), float(scale), p, dptr(ws), nws, stream_ptr(x.device)), "mi355_mhsa_fwd")
    return y


def sdpa16(qkv16, num_heads, scale, precision=None):
    qkv16 = _require16(qkv16, "qkv16", precision)
    B, N, C3 = qkv16.shape
    C = C3 // 3
    out = torch.empty(B, N, C, dtype=qkv16.dtype, device=qkv16.device)
    check(lib().mi355_sdpa16_fwd(dptr(qkv16), dptr(out), B, N, num_heads, C // num_heads, float(scale), _prec(precision),
                                 stream_ptr(qkv16.device)), "mi355_sdpa16_fwd")
    return out


def cswin_lepe_attention16(qkv16, getv_w, getv_b, out16, reso, c0, Cb, heads, Hsp, Wsp, scale, precision=None):
    qkv16 = _require16(qkv16, "qkv16", precision)
    getv_w = require_device_f32(getv_w, "get_v.weight")
    getv_b = require_device_f32(getv_b, "get_v.bias")
    B = qkv16.shape[0]
    Ctot = qkv16.shape[-1] // 3
    check(lib().mi355_cswin_lepe_attn16_fwd(dptr(qkv16), dptr(getv_w), dptr(getv_b), dptr(out16), B, reso, Ctot, c0, Cb, heads,
                                            Hsp, Wsp, float(scale), _prec(precision), stream_ptr(qkv16.device)),
          "mi355_cswin_lepe_attn16_fwd")
    return out16


# ---- attention cores ----------------------------------------------------------------------------------------
def sdpa(qkv, num_heads, scale, precision=None):
    """qkv (B,N,3*C) straight from the qkv Linear -> (B,N,C) = concat_heads(softmax(QK^T*scale) V)."""
    qkv = require_device_f32(qkv, "qkv")
    B, N, C3 = qkv.shape
    C = C3 // 3
    d = C // num_heads
    out = torch.empty(B, N, C, dtype=torch.float32, device=qkv.device)
    check(lib().mi355_sdpa_fwd(dptr(qkv), dptr(out), B, N, num_heads, d, float(scale), _prec(precision),
                               stream_ptr(qkv.device)), "mi355_sdpa_fwd")
    return out


def cswin_lepe_attention(qkv, getv_w, getv_b, out, reso, c0, Cb, heads, Hsp, Wsp, scale, precision=None):
    """One LePEAttention branch on the channel slice [c0, c0+Cb) of a (B,L,3,Ctot) qkv buffer; writes `out` (B,L,Ctot)."""
    qkv = require_device_f32(qkv, "qkv")
    getv_w = require_device_f32(getv_w, "get_v.weight")
    getv_b = require_device_f32(getv_b, "get_v.bias")
    B, L = qkv.shape[0], qkv.shape[1]
    Ctot = qkv.shape[-1] // 3 if qkv.dim() == 3 else qkv.shape[-1]
    check(lib().mi355_cswin_lepe_attn_fwd(dptr(qkv), dptr(getv_w), dptr(getv_b), dptr(out), B, reso, Ctot, c0, Cb,
                                          heads, Hsp, Wsp, float(scale), _prec(precision), stream_ptr(qkv.device)),
          "mi355_cswin_lepe_attn_fwd")
    return out


def xca_core(qkv, temperature, num_heads, precision=None, out16=False):
    """`out16`: the context in the 16-bit operand format of `precision` (what linear16 reads) instead of fp32; qkv may then be in that
    format as well (the 16-bit output of the qkv GEMM)."""
    temperature = require_device_f32(temperature, "temperature").reshape(-1)
    B, N, C3 = qkv.shape
    C = C3 // 3
    if out16:
        is16 = qkv.dtype == dtype16(precision)
        if not is16:
            qkv = require_device_f32(qkv, "qkv")
        elif not (qkv.is_cuda and qkv.is_contiguous()):
            raise ValueError("xca_core: 16-bit qkv must be a contiguous device tensor")
        out = torch.empty(B, N, C, dtype=dtype16(precision), device=qkv.device)
        check(lib().mi355_xca16_fwd(dptr(qkv), 1 if is16 else 0, dptr(temperature), dptr(out), B, N, num_heads, C // num_heads,
                                    _prec(precision), stream_ptr(qkv.device)), "mi355_xca16_fwd")
        return out
    qkv = require_device_f32(qkv, "qkv")
    out = torch.empty(B, N, C, dtype=torch.float32, device=qkv.device)
    check(lib().mi355_xca_fwd(dptr(qkv), dptr(temperature), dptr(out), B, N, num_heads, C // num_heads,
                              _prec(precision), stream_ptr(qkv.device)), "mi355_xca_fwd")
    return out


def lpi(x, w1, b1, bn_w, bn_b, bn_mean, bn_var, bn_eps, w2, b2, H, W, gamma=None, resid=None, ln=None, stats=None):
    """XCiT LPI on tokens x (B,N,C) with eval-mode BatchNorm; optional fused `resid + gamma * LPI(x)`.  `ln` (an nn.LayerNorm): the
    block becomes resid + gamma * LPI(ln(x)) with the normalisation applied on the way into the stencil kernel (mi355_ln_lpi_fwd)."""
    x = require_device_f32(x, "x")
    B, N, C = x.shape
    if N != H * W:
        raise ValueError(f"lpi: {N} tokens do not form a {H}x{W} grid")
    pre = [require_device_f32(t, n) for t, n in ((w1, "conv1.weight"), (b1, "conv1.bias"), (bn_w, "bn.weight"),
                                                 (bn_b, "bn.bias"), (bn_mean, "bn.running_mean"),
                                                 (bn_var, "bn.running_var"))]
    post = [require_device_f32(t, n) for t, n in ((w2, "conv2.weight"), (b2, "conv2.bias"))]
    gamma, resid = _opt(gamma, "gamma"), _opt(resid, "resid")
    y = torch.empty_like(x)
    n = lib().mi355_lpi_workspace_bytes(B, H, W, C)
    ws = workspace(n, x.device)
    if ln is not None and stats is not None:              # `stats`: (mean, rstd) per token, written by the GEMM that produced x (linear16_stats)
        lw, lb = require_device_f32(ln.weight, "ln.weight"), require_device_f32(ln.bias, "ln.bias")
        stats = require_device_f32(stats, "stats")
        if stats.numel() != B * N * 2:
            raise ValueError("lpi: stats must hold (mean, rstd) for every token")
        check(lib().mi355_ln_lpi_stats_fwd(dptr(x), dptr(stats), dptr(lw), dptr(lb), *[dptr(a) for a in pre], float(bn_eps),
                                           *[dptr(a) for a in post], dptr(gamma), dptr(resid), dptr(y), B, H, W, C, stream_ptr(x.device)),
              "mi355_ln_lpi_stats_fwd")
        return y
    if ln is not None:
        lw, lb = require_device_f32(ln.weight, "ln.weight"), require_device_f32(ln.bias, "ln.bias")
        check(lib().mi355_ln_lpi_fwd(dptr(x), dptr(lw), dptr(lb), float(ln.eps), *[dptr(a) for a in pre], float(bn_eps),
                                     *[dptr(a) for a in post], dptr(gamma), dptr(resid), dptr(y), B, H, W, C, dptr(ws), ws.numel(),
                                     stream_ptr(x.device)), "mi355_ln_lpi_fwd")
        return y
    check(lib().mi355_lpi_fwd(dptr(x), *[dptr(a) for a in pre], float(bn_eps), *[dptr(a) for a in post], dptr(gamma),
                              dptr(resid), dptr(y), B, H, W, C, dptr(ws), ws.numel(), stream_ptr(x.device)),
          "mi355_lpi_fwd")
    return y


def patch_embed(img, wp, bp, cls, pos, patch, precision=None):
    img = require_device_f32(img, "img")
    wp = require_device_f32(wp, "proj.weight")
    E = wp.shape[0]
    wp = wp.reshape(E, -1)
    bp = require_device_f32(bp, "proj.bias")
    cls, pos = _opt(cls, "cls_token"), _opt(pos, "pos")
    B, Cin, H, W = img.shape
    P = (H // patch) * (W // patch)
    if (cls is None) != (pos is None):
        raise ValueError("patch_embed: pass both cls and pos (ViT) or neither (plain patches)")
    if pos is not None and pos.numel() != (P + 1) * E:
        raise ValueError("patch_embed: position embedding does not match the patch grid")
    tokens = torch.empty(B, P + (1 if cls is not None else 0), E, dtype=torch.float32, device=img.device)
    n = lib().mi355_patch_embed_workspace_bytes(B, Cin, H, W, patch, E, _prec(precision)) if cls is not None else 0
    if n:
        ws = _ffi.workspace_named("patch_embed", n, img.device)
        check(lib().mi355_patch_embed_ws_fwd(dptr(img), dptr(wp), dptr(bp), dptr(cls), dptr(pos), dptr(tokens), B, Cin, H, W,
                                             patch, E, _prec(precision), dptr(ws), ws.numel(), stream_ptr(img.device)),
              "mi355_patch_embed_ws_fwd")
        return tokens
    check(lib().mi355_patch_embed_fwd(dptr(img), dptr(wp), dptr(bp), dptr(cls), dptr(pos), dptr(tokens), B, Cin, H, W,
                                      patch, E, _prec(precision), stream_ptr(img.device)), "mi355_patch_embed_fwd")
    return tokens


def _gemm_rows(w, in_layout):
    cout = w.shape[0]
    rows = (w if in_layout == 0 else w.permute(0, 2, 3, 1)).reshape(cout, -1)
    k = rows.shape[1]
    out = torch.zeros(cout, (k + 3) // 4 * 4, dtype=torch.float32, device=w.device)
    out[:, :k] = rows
    return out


def conv_weight_rows(param, in_layout):
    """Weight of a Conv2d as the GEMM operand of mi355_conv2d_tokens_fwd: (Cout, ldw) rows, zero padded to ldw % 4 == 0,
    K ordered (c,ky,kx) for NCHW input (in_layout 0) or (ky,kx,c) for token-major input (in_layout 1).  A parameter layout
    transform done once per parameter version (like weight16)."""
    tag = (param._version, param.data_ptr(), tuple(param.shape))
    return _derived_get((param,), ("rows", in_layout), tag, lambda: _gemm_rows(param.detach(), in_layout))


def _bn_tag(bn):
    return tuple((t._version, t.data_ptr()) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (bn.eps,)


def conv_bn_rows(conv_w, bn, in_layout):
    """Conv weight with the eval-mode BatchNorm that follows it folded in (xcit.py:79-86 conv3x3 = Conv2d(bias=False) + BatchNorm2d):
    w' = w * s, b' = beta - mean * s with s = gamma / sqrt(var + eps), as GEMM rows like conv_weight_rows.  Cached per version
    of the five tensors involved."""
    def build():
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        bias = (bn.bias.detach() - bn.running_mean.detach() * s).contiguous()
        return _gemm_rows(conv_w.detach() * s[:, None, None, None], in_layout), bias

    tag = ((conv_w._version, conv_w.data_ptr(), tuple(conv_w.shape)),) + _bn_tag(bn)
    return _derived_get((conv_w, bn), ("bnrows", in_layout), tag, build)


def conv2d_tokens(x, weight, bias, kernel, stride, pad, in_layout, hw=None, precision=None, act=ACT_NONE, pos=None, wrows=None):
    """Conv2d -> token-major (B, OH*OW, Cout) = act(conv + bias + pos).  x is NCHW (in_layout 0) or tokens (B, H*W, Cin) with
    hw=(H, W) (in_layout 1).  `wrows` passes prepared GEMM rows (conv_bn_rows) instead of a Conv2d weight."""
    x = require_device_f32(x, "x")
    if in_layout == 0:
        B, Cin, H, W = x.shape
    else:
        B, L, Cin = x.shape
        H, W = hw
        if L != H * W:
            raise ValueError("conv2d_tokens: token count does not match hw")
    if wrows is None:
        wrows = conv_weight_rows(weight, in_layout)
    Cout = wrows.shape[0]
    bias = _opt(bias, "bias")
    OH = (H + 2 * pad - kernel) // stride + 1
    OW = (W + 2 * pad - kernel) // stride + 1
    if pos is not None:
        pos = require_device_f32(pos, "pos")
        if pos.numel() != OH * OW * Cout:
            raise ValueError("conv2d_tokens: pos must hold one row per output token")
    y = torch.empty(B, OH * OW, Cout, dtype=torch.float32, device=x.device)
    check(lib().mi355_conv2d_tokens_fwd(dptr(x), dptr(wrows), dptr(bias), dptr(pos), dptr(y), B, Cin, H, W, Cout, kernel, kernel,
                                        stride, pad, wrows.shape[1], in_layout, act, _prec(precision), stream_ptr(x.device)),
          "mi355_conv2d_tokens_fwd")
    return y, (OH, OW)


def _rows3(t, name):
    """(B, N, C') device view with unit channel stride and a dense batch axis -> (tensor, row stride)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dim() != 3:
        raise _ffi.Mi355Error(f"{name}: expected a (B, N, C) device tensor (there is no CPU path)")
    if t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
        t = t.contiguous()
    return t, t.stride(1)


def sdpa_general(q, k, v, num_heads, scale, bias=None, precision=None, out=None):
    """softmax(q k^T * scale + bias) v for (B,Nq,C) queries and (B,Nkv,C) keys / values (views into fused projections welcome).
    fp32 tensors -> fp32 result; fp16 / bf16 tensors (the operand type of `precision`) -> same type.  bias: (heads,Nq,Nkv) or
    (B,heads,Nq,Nkv) fp32.  out: optional (B,Nq,C) destination view with unit channel stride and a dense batch axis (a channel
    slice of a wider tensor: the C entry takes the row stride)."""
    p = _prec(precision)
    q, ldq = _rows3(q, "q")
    k, ldk = _rows3(k, "k")
    v, ldv = _rows3(v, "v")
    B, Nq, C = q.shape
    Nkv = k.shape[1]
    if k.shape != (B, Nkv, C) or v.shape != (B, Nkv, C) or C % num_heads:
        raise ValueError("sdpa_general: q (B,Nq,C), k/v (B,Nkv,C) with C divisible by num_heads")
    io16 = q.dtype != torch.float32
    if io16 and (p == PREC_STRICT or q.dtype != dtype16(p)):
        raise ValueError("sdpa_general: 16-bit tensors must be in the operand type of the precision mode")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("sdpa_general: q, k, v must share one element type")
    bstride = 0
    if bias is not None:
        bias = require_device_f32(bias, "bias")
        if bias.dim() == 4 and bias.shape[0] == B and B > 1:
            bstride = num_heads * Nq * Nkv
        if bias.numel() != (B if bstride else 1) * num_heads * Nq * Nkv:
            raise ValueError("sdpa_general: bias must be (heads,Nq,Nkv) or (B,heads,Nq,Nkv)")
    if out is None:
        out = torch.empty(B, Nq, C, dtype=q.dtype, device=q.device)
    elif (tuple(out.shape) != (B, Nq, C) or out.dtype != q.dtype or out.device != q.device or out.stride(2) != 1 or
          out.stride(0) != Nq * out.stride(1)):
        raise ValueError("sdpa_general: out must be a (B,Nq,C) view of q's type with unit channel stride and a dense batch axis")
    check(lib().mi355_sdpa_general_fwd(dptr(q), dptr(k), dptr(v), dptr(bias), dptr(out), B, num_heads, Nq, Nkv, C // num_heads,
                                       ldq, ldk, ldv, out.stride(1), bstride, float(scale), 1 if io16 else 0, p, stream_ptr(q.device)),
          "mi355_sdpa_general_fwd")
    return out


def tokens_to_nchw(tokens, H, W):
    """(B, H*W, C) -> (B, C, H, W)."""
    tokens = require_device_f32(tokens, "tokens")
    B, L, C = tokens.shape
    if L != H * W:
        raise ValueError("tokens_to_nchw: token count does not match (H, W)")
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=tokens.device)
    check(lib().mi355_tokens_to_nchw_axpy_fwd(dptr(tokens), dptr(None), dptr(None), dptr(y), B, L, C, stream_ptr(tokens.device)),
          "mi355_tokens_to_nchw_axpy_fwd")
    return y


def dwconv_bn_nchw_tokens(x, conv_w, conv_b, bn):
    """Depth-wise conv (stride 1, 'same' padding) + eval BatchNorm2d of an NCHW map, written token-major (B, H*W, C); conv bias and
    BatchNorm are folded into one weight / bias pair, cached per version."""
    x = require_device_f32(x, "x")
    B, C, H, W = x.shape
    ks = conv_w.shape[-1]

    def build():
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        cb = conv_b.detach() if conv_b is not None else torch.zeros_like(s)
        return (conv_w.detach().reshape(C, ks * ks) * s[:, None]).contiguous(), ((cb - bn.running_mean.detach()) * s + bn.bias.detach()).contiguous()

    tag = ((conv_w._version, conv_w.data_ptr()),) + _bn_tag(bn) + ((conv_b._version, conv_b.data_ptr()) if conv_b is not None else ())
    w, b = _derived_get((conv_w, bn), ("dwbn_nchw",), tag, build)
    y = torch.empty(B, H * W, C, dtype=torch.float32, device=x.device)
    check(lib().mi355_dwconv_nchw_tokens_fwd(dptr(x), dptr(w), dptr(b), dptr(y), B, C, H, W, ks, stream_ptr(x.device)),
          "mi355_dwconv_nchw_tokens_fwd")
    return y


def pooled_pyramid_tokens(x, H, W, sizes, d_convs):
    """P2T's key/value source (p2t.py:76-82): for every (oh, ow) in `sizes`, adaptive-average-pool the (H x W) token grid x (B, H*W, C),
    add the depth-wise 3x3 conv of the pooled grid (`d_convs[i]`, a Conv2d with groups == C) and concatenate along the token axis."""
    x = require_device_f32(x, "x")
    B, L, C = x.shape
    if L != H * W:
        raise ValueError("pooled_pyramid_tokens: token count does not match (H, W)")
    total = sum(oh * ow for oh, ow in sizes)
    out = torch.empty(B, total, C, dtype=torch.float32, device=x.device)
    off = 0
    for (oh, ow), conv in zip(sizes, d_convs):
        if conv.kernel_size != (3, 3) or conv.groups != C or conv.stride != (1, 1) or conv.padding != (1, 1):
            raise ValueError("pooled_pyramid_tokens: d_convs must be depth-wise 3x3, stride 1, padding 1")
        pooled = torch.empty(B, oh * ow, C, dtype=torch.float32, device=x.device)
        check(lib().mi355_adaptive_pool_tokens_fwd(dptr(x), dptr(pooled), B, H, W, C, oh, ow, stream_ptr(x.device)),
              "mi355_adaptive_pool_tokens_fwd")
        w = require_device_f32(conv.weight, "d_conv.weight").reshape(C, 9)
        b = _opt(conv.bias, "d_conv.bias")
        dst = out[:, off:off + oh * ow]
        check(lib().mi355_dwconv3x3_tokens_residual_fwd(dptr(pooled), dptr(w), dptr(b), ctypes.c_void_p(dst.data_ptr()), B, oh, ow, C,
                                                        total * C, stream_ptr(x.device)), "mi355_dwconv3x3_tokens_residual_fwd")
        off += oh * ow
    return out


def qk_logits(q, k, num_heads, precision=PREC_STRICT):
    """Unscaled logits (B, heads, Nq, Nkv) of (B,Nq,C) queries against (B,Nkv,C) keys (views into fused projections welcome)."""
    q, ldq = _rows3(q, "q")
    k, ldk = _rows3(k, "k")
    B, Nq, C = q.shape
    Nkv = k.shape[1]
    if q.dtype != torch.float32 or k.dtype != torch.float32:
        raise TypeError("qk_logits: fp32 tensors expected")
    out = torch.empty(B, num_heads, Nq, Nkv, dtype=torch.float32, device=q.device)
    check(lib().mi355_qk_logits_fwd(dptr(q), dptr(k), dptr(out), B, num_heads, Nq, Nkv, C // num_heads, ldq, ldk, _prec(precision),
                                    stream_ptr(q.device)), "mi355_qk_logits_fwd")
    return out


def topk_mask_(logits, k):
    """In place: the k largest entries of every last-axis row -> 0, the others -> -1e30 (an additive attention bias)."""
    logits = require_device_f32(logits, "logits")
    N = logits.shape[-1]
    check(lib().mi355_topk_mask_fwd(dptr(logits), logits.numel() // N, N, int(k), stream_ptr(logits.device)), "mi355_topk_mask_fwd")
    return logits


def head_padded(weight, bias, groups, d, dp, axis):
    """Linear parameters with every head's slice of width d padded to dp with zeros -- along the output features (axis 0: weight rows
    and bias, `groups` = number of head slices) or the input features (axis 1: weight columns).  Lets head widths outside the
    attention kernel's {32, 64} run on it: zero q/k columns add nothing to the logits, zero v columns produce zero outputs that the
    padded projection ignores.  Cached with the parameters."""
    def build():
        w = weight.detach()
        if axis == 0:
            wp = torch.zeros(groups * dp, w.shape[1], dtype=torch.float32, device=w.device)
            wp.view(groups, dp, -1)[:, :d] = w.view(groups, d, -1)
            bp = None
            if bias is not None:
                bp = torch.zeros(groups * dp, dtype=torch.float32, device=w.device)
                bp.view(groups, dp)[:, :d] = bias.detach().view(groups, d)
            return wp, bp
        wp = torch.zeros(w.shape[0], groups * dp, dtype=torch.float32, device=w.device)
        wp.view(-1, groups, dp)[:, :, :d] = w.view(-1, groups, d)
        return wp, None

    anchors = (weight,) if bias is None else (weight, bias)
    tag = tuple((t._version, t.data_ptr()) for t in anchors)
    return _derived_get(anchors, ("headpad", groups, d, dp, axis), tag, build)


SDPA_WIDTHS = (32, 64, 128, 192, 256)      # head widths mi355_sdpa_general_fwd is built for


def attn_head_width(d):
    """Head width the streaming attention kernel runs a logical width d on (zero padded)."""
    for w in SDPA_WIDTHS:
        if d <= w:
            return w
    raise ValueError(f"attention head width {d} > {SDPA_WIDTHS[-1]} is outside the kernel's envelope")


def vit_pos_table(position_embedding, patch_size, H, W):
    """Rows to add to the tokens of a ViT forward at image size (H, W): the parameter itself at the native resolution, otherwise
    ViT.py:160-178 -- row 0 kept, rows 1.. resized bicubically from their (n0, n0) grid to (W // patch, H // patch) with the
    reference's scale factors ((W // patch + 0.1) / n0, (H // patch + 0.1) / n0).  Cached per (H, W) and parameter version."""
    import math
    pe = position_embedding
    N = pe.shape[1] - 1
    E = pe.shape[2]
    n_tok = (H // patch_size) * (W // patch_size)
    if n_tok == N and W == H:
        return pe.reshape(N + 1, E)
    n0 = int(math.sqrt(N))
    w0, h0 = W // patch_size, H // patch_size
    if n0 * n0 != N:
        raise ValueError("position embedding does not hold a square patch grid")

    def build():
        src = require_device_f32(pe.detach().reshape(N + 1, E), "position_embedding")
        out = torch.empty(1 + w0 * h0, E, dtype=torch.float32, device=src.device)
        axpby(src, out, 1, E, E, E)                                                       # row 0 as is
        check(lib().mi355_bicubic_rows_fwd(ctypes.c_void_p(src.data_ptr() + E * 4), ctypes.c_void_p(out.data_ptr() + E * 4), n0, n0,
                                           w0, h0, E, (w0 + 0.1) / math.sqrt(N), (h0 + 0.1) / math.sqrt(N), stream_ptr(src.device)),
              "mi355_bicubic_rows_fwd")
        return out

    return _derived_get((pe,), ("vitpos", patch_size, H, W), (pe._version, pe.data_ptr()), build)


def dwconv_patch_tokens(x, conv_w, conv_b, bn, H, W, sr):
    """Depth-wise conv (kernel == stride == sr, optional bias) + optional eval BatchNorm2d on a token grid:
    (B, H*W, C) -> (B, (H/sr)*(W/sr), C).  Conv bias and BatchNorm are folded into one weight / bias pair, cached per version."""
    x = require_device_f32(x, "x")
    B, L, C = x.shape
    if L != H * W:
        raise ValueError("dwconv_patch_tokens: token count does not match (H, W)")
    def build():
        w = conv_w.detach().reshape(C, sr * sr)
        cb = None if conv_b is None else conv_b.detach()
        if bn is None:
            return w.contiguous(), cb
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        shift = bn.running_mean.detach() if cb is None else bn.running_mean.detach() - cb
        return (w * s[:, None]).contiguous(), (bn.bias.detach() - shift * s).contiguous()

    tag = ((conv_w._version, conv_w.data_ptr(), tuple(conv_w.shape)),) + \
          (() if conv_b is None else ((conv_b._version, conv_b.data_ptr()),)) + (() if bn is None else _bn_tag(bn))
    anchors = (conv_w,) + (() if conv_b is None else (conv_b,)) + (() if bn is None else (bn,))
    wf, bf = _derived_get(anchors, ("dwpatch", sr), tag, build)
    y = torch.empty(B, (H // sr) * (W // sr), C, dtype=torch.float32, device=x.device)
    check(lib().mi355_dwconv_patch_tokens_fwd(dptr(x), dptr(wf), dptr(bf), dptr(y), B, H, W, C, sr, stream_ptr(x.device)),
          "mi355_dwconv_patch_tokens_fwd")
    return y


def class_attention(q, k, v, num_heads, scale, N, ldq, ldkv):
    """One query per (image, head) over N keys; q/k/v are (possibly strided) views into fp32 device tensors."""
    B = q.shape[0]
    C = q.shape[-1]
    out = torch.empty(B, C, dtype=torch.float32, device=q.device)
    check(lib().mi355_class_attn_fwd(dptr(q), dptr(k), dptr(v), dptr(out), B, N, num_heads, C // num_heads, ldq, ldkv, float(scale),
                                     stream_ptr(q.device)), "mi355_class_attn_fwd")
    return out


def axpby(x, y, rows, cols, ldx, ldy, alpha=1.0, u=None, ldu=0, gamma=None):
    """y[r,c] = alpha*x[r,c] + gamma[c]*u[r,c] on strided row views (pointers = the tensors' data_ptr, strides in floats)."""
    for t in (x, y, u, gamma):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise _ffi.Mi355Error("axpby: fp32 device tensors only (there is no CPU path)")
    check(lib().mi355_axpby_fwd(dptr(x), dptr(u), dptr(gamma), dptr(y), rows, cols, ldx, ldu, ldy, float(alpha),
                                stream_ptr(x.device)), "mi355_axpby_fwd")
    return y


def token_mean(x, skip_first=0):
    """Mean over the token axis of x (B,N,C), optionally skipping the first `skip_first` tokens."""
    x = require_device_f32(x, "x")
    B, N, C = x.shape
    y = torch.empty(B, C, dtype=torch.float32, device=x.device)
    base = ctypes.c_void_p(x.data_ptr() + skip_first * C * 4)
    check(lib().mi355_token_mean_fwd(base, dptr(y), B, N - skip_first, C, N * C, stream_ptr(x.device)), "mi355_token_mean_fwd")
    return y


def stream_copy(src, dst):
    check(lib().mi355_stream_copy(dptr(src), dptr(dst), src.numel() * src.element_size(), stream_ptr(src.device)),
          "mi355_stream_copy")
    return dst


def mfma_yardstick(device, shape=0, target_ms=40.0):
    """Box calibration (include/mi355attn.h mi355_mfma_yardstick): a register-operand MFMA loop on every SIMD of `device`, sized to run
    about `target_ms`.  Returns dict(TFLOPs, ms, sclk_MHz_counter = the wave's cycle counter against the 100 MHz wall clock,
    sclk_MHz_issue = clock implied by the instruction's documented issue interval (32 cycles per 32x32x16, shape 1 only), mfma, flop)."""
    from ._ffi import StreamTimer
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    rep = torch.zeros(4, dtype=torch.int64, device=device)
    per_iter, flop_per = (8, 16384) if shape == 0 else (4, 32768)

    def run(iters):
        tm = StreamTimer(device)
        tm.start()
        check(lib().mi355_mfma_yardstick(shape, iters, dptr(sink), dptr(rep), stream_ptr(device)), "mi355_mfma_yardstick")
        return tm.stop_ms()

    run(256)                                                 # code object load + a first look at the rate
    ms = run(4096)
    iters = int(min(1 << 24, max(4096, 4096 * target_ms / max(ms, 1e-3))))
    ms = run(iters)
    ticks, ref, groups = (int(v) for v in rep[:3].tolist())
    mfma = groups * 4 * iters * per_iter
    out = {"TFLOPs": round(mfma * flop_per / (ms * 1e-3) / 1e12, 1), "ms": round(ms, 2), "mfma": mfma, "flop": mfma * flop_per,
           "sclk_MHz_counter": round(ticks / ref * 100.0, 1) if ref else None}
    if shape == 1 and groups:
        per_simd = 2 * iters * per_iter                      # two waves per SIMD (two 4-wave workgroups per CU)
        out["sclk_MHz_issue"] = round(per_simd * 32 / (ms * 1e-3) / 1e6, 1)
    return out
