"""ctypes binding of libmi355attn.so (the C ABI declared in include/mi355attn.h).

The library is loaded lazily at the first op call.  There is NO fallback: if the shared object is missing
or the input is not a CUDA(HIP) tensor the call raises -- the product path never routes through a CPU
implementation (see DESIGN.md "boundary").

`import torch` happens before the CDLL load on purpose: torch's wheel bundles libamdhip64.so (same SONAME
as /opt/rocm's), so loading torch first makes our library bind to the runtime torch already initialised --
one HIP runtime per process, shared streams and device pointers.
"""
import collections
import ctypes
import os
import threading
import warnings

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355attn.so")
ABI_VERSION = 1

_lib = None
_lock = threading.Lock()

c_f32p = ctypes.c_void_p      # device pointers travel as void*
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_vp = ctypes.c_void_p
c_float = ctypes.c_float

# name -> (restype, argtypes); mirrors include/mi355attn.h one-to-one (tests/test_abi.py checks the match)
SIGNATURES = {
    "mi355_version": (c_int, []),
    "mi355_last_error": (ctypes.c_char_p, []),
    "mi355_set_option": (c_int, [ctypes.c_char_p, ctypes.c_long]),
    "mi355_set_default_option": (c_int, [ctypes.c_char_p, ctypes.c_long]),
    "mi355_get_option": (ctypes.c_long, [ctypes.c_char_p]),
    "mi355_workspace_forget": (c_int, [c_vp, ctypes.c_size_t]),
    "mi355_trace_begin": (c_int, []),
    "mi355_trace_end": (ctypes.c_long, [ctypes.c_char_p, c_size]),
    "mi355_sync_status": (c_int, []),
    "mi355_range_status": (c_int, []),
    "mi355_range_arm": (c_int, [c_int]),
    "mi355_range_wait": (c_int, []),
    "mi355_range_launches": (ctypes.c_long, []),
    "mi355_se_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_se_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "mi355_se_ex_fwd": (c_int, [c_vp] * 6 + [c_int] * 6 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_eca_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_eca_fwd": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "mi355_cbam_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_cbam_fwd": (c_int, [c_vp] * 5 + [c_int] * 7 + [c_vp, c_size, c_vp]),
    "mi355_double_attn_workspace_bytes": (c_size, [c_int] * 6),
    "mi355_double_attn_ws_bytes": (c_size, [c_int] * 7),
    "mi355_double_attn_fwd": (c_int, [c_vp] * 10 + [c_int] * 7 + [c_vp, c_size, c_vp]),
    "mi355_linear_fwd": (c_int, [c_vp] * 6 + [c_int] * 7 + [c_vp]),
    "mi355_token_mix_fwd": (c_int, [c_vp] * 5 + [c_int] * 6 + [c_vp]),
    "mi355_layernorm_fwd": (c_int, [c_vp] * 4 + [c_int, c_int, c_float, c_vp]),
    "mi355_sdpa_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, c_int, c_vp]),
    "mi355_cswin_lepe_attn_fwd": (c_int, [c_vp] * 4 + [c_int] * 8 + [c_float, c_int, c_vp]),
    "mi355_xca_fwd": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp]),
    "mi355_xca16_fwd": (c_int, [c_vp, c_int, c_vp, c_vp] + [c_int] * 5 + [c_vp]),
    "mi355_lpi_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_lpi_fwd": (c_int, [c_vp] * 7 + [c_float] + [c_vp] * 5 + [c_int] * 4 + [c_vp, c_size, c_vp]),
    "mi355_ln_lpi_fwd": (c_int, [c_vp] * 3 + [c_float] + [c_vp] * 6 + [c_float] + [c_vp] * 5 + [c_int] * 4 + [c_vp, c_size, c_vp]),
    "mi355_patch_embed_fwd": (c_int, [c_vp] * 6 + [c_int] * 7 + [c_vp]),
    "mi355_patch_embed_workspace_bytes": (c_size, [c_int] * 7),
    "mi355_patch_embed_ws_fwd": (c_int, [c_vp] * 6 + [c_int] * 7 + [c_vp, c_size, c_vp]),
    "mi355_cast16_fwd": (c_int, [c_vp, c_vp, c_size, c_int, c_vp]),
    "mi355_layernorm16_fwd": (c_int, [c_vp] * 4 + [c_int, c_int, c_float, c_int, c_vp]),
    "mi355_ln_linear16_fwd": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_float, c_int, c_int, c_int, c_vp]),
    "mi355_layernorm16_t_fwd": (c_int, [c_vp] * 4 + [c_int] * 4 + [c_float, c_int, c_vp]),
    "mi355_mixer_token_workspace_bytes": (c_size, [c_int] * 3),
    "mi355_mixer_token_fwd": (c_int, [c_vp] * 3 + [c_float] + [c_vp] * 5 + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "mi355_linear16_tr_fwd": (c_int, [c_vp] * 5 + [c_int] * 6 + [c_vp]),
    "mi355_linear16_fwd": (c_int, [c_vp] * 6 + [c_int] * 8 + [c_vp]),
    "mi355_linear16_workspace_bytes": (c_size, [c_int] * 3),
    "mi355_linear16_ws_fwd": (c_int, [c_vp] * 6 + [c_int] * 8 + [c_vp, c_size, c_vp]),
    "mi355_sdpa16_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, c_int, c_vp]),
    "mi355_cswin_lepe_attn16_fwd": (c_int, [c_vp] * 4 + [c_int] * 8 + [c_float, c_int, c_vp]),
    "mi355_conv2d_tokens_fwd": (c_int, [c_vp] * 5 + [c_int] * 13 + [c_vp]),
    "mi355_token_mean_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, ctypes.c_long, c_vp]),
    "mi355_chan_stat_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "mi355_simam_fwd": (c_int, [c_vp, c_vp] + [c_int] * 4 + [ctypes.c_float, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_srm_fwd": (c_int, [c_vp] * 6 + [ctypes.c_float, c_vp] + [c_int] * 4 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_gct_gauss_fwd": (c_int, [c_vp, c_vp] + [c_int] * 4 + [ctypes.c_float, ctypes.c_float, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_lct_fwd": (c_int, [c_vp] * 4 + [c_int] * 5 + [ctypes.c_float, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_gct_fwd": (c_int, [c_vp] * 5 + [c_int] * 4 + [ctypes.c_float, c_int, c_int, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_axis_attn_workspace_bytes": (ctypes.c_size_t, [c_int] * 4),
    "mi355_gc_fwd": (c_int, [c_vp] * 10 + [c_int] * 5 + [ctypes.c_float, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_coordatt_fwd": (c_int, [c_vp] * 10 + [c_int] * 5 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_triplet_fwd": (c_int, [c_vp] * 6 + [c_int] * 5 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_bam_workspace_bytes": (ctypes.c_size_t, [c_int] * 5),
    "mi355_bam_fwd": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_bam_gates_fwd": (c_int, [c_vp] * 4 + [c_int] * 6 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_zpool_fwd": (c_int, [c_vp, c_vp] + [c_int] * 4 + [c_vp]),
    "mi355_attention_gate_workspace_bytes": (ctypes.c_size_t, [c_int] * 3),
    "mi355_attention_gate_fwd": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_ln_fold_stats_bytes": (ctypes.c_size_t, [c_int] * 2),
    "mi355_ln_center16_fwd": (c_int, [c_vp] * 4 + [c_int, c_int, c_float, c_int, c_vp]),
    "mi355_ln_finalize_fwd": (c_int, [c_vp] * 5 + [c_int, c_int, c_float, c_float, c_int, c_vp, c_vp]),
    "mi355_linear16_emit_fwd": (c_int, [c_vp] * 5 + [c_int] * 6 + [c_vp] * 4),
    "mi355_linear16_lnfold_fwd": (c_int, [c_vp] * 6 + [c_int] * 7 + [c_vp]),
    "mi355_mhsa_workspace_bytes": (ctypes.c_size_t, [c_int] * 4),
    "mi355_mhsa_fwd": (c_int, [c_vp, c_int] + [c_vp] * 6 + [c_int] * 4 + [c_float, c_int, c_vp, ctypes.c_size_t, c_vp]),
    "mi355_sk_workspace_bytes": (ctypes.c_size_t, [c_int] * 4),
    "mi355_sk_fwd": (c_int, [c_vp] * 3 + [c_int] * 7 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_cam_workspace_bytes": (ctypes.c_size_t, [c_int] * 2),
    "mi355_cam_fwd": (c_int, [c_vp] * 3 + [c_int] * 5 + [c_vp, ctypes.c_size_t, c_vp]),
    "mi355_tokens_to_nchw_axpy_fwd": (c_int, [c_vp] * 4 + [c_int] * 3 + [c_vp]),
    "mi355_dwconv_nchw_tokens_fwd": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_vp]),
    "mi355_qk_logits_fwd": (c_int, [c_vp] * 3 + [c_int] * 8 + [c_vp]),
    "mi355_topk_mask_fwd": (c_int, [c_vp, ctypes.c_long, c_int, c_int, c_vp]),
    "mi355_adaptive_pool_tokens_fwd": (c_int, [c_vp] * 2 + [c_int] * 6 + [c_vp]),
    "mi355_dwconv3x3_tokens_residual_fwd": (c_int, [c_vp] * 4 + [c_int] * 4 + [ctypes.c_long, c_vp]),
    "mi355_mlp_fused_fwd": (c_int, [c_vp] * 7 + [ctypes.c_long, c_int, c_int, c_int, ctypes.c_float, c_int, c_vp]),
    "mi355_proj_mlp_fused_fwd": (c_int, [c_vp] * 10 + [ctypes.c_long, c_int, c_int, c_int, ctypes.c_float, c_int, c_vp]),
    "mi355_sdpa_general_fwd": (c_int, [c_vp] * 5 + [c_int] * 5 + [ctypes.c_long] * 5 + [ctypes.c_float, c_int, c_int, c_vp]),
    "mi355_dwconv_patch_tokens_fwd": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_vp]),
    "mi355_cswin_lepe_attn16_pair_fwd": (c_int, [c_vp] * 6 + [c_int] * 5 + [ctypes.c_float, c_int, c_vp]),
    "mi355_cswin_stripe_attn_fwd": (c_int, [c_vp] * 8 + [c_int] * 5 + [c_float, c_float, c_int, c_vp]),
    "mi355_class_attn_fwd": (c_int, [c_vp] * 4 + [c_int] * 4 + [ctypes.c_long, ctypes.c_long, ctypes.c_float, c_vp]),
    "mi355_axpby_fwd": (c_int, [c_vp] * 4 + [ctypes.c_long, c_int, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_float, c_vp]),
    "mi355_comm_unique_id": (c_int, [c_vp, c_size]),
    "mi355_comm_init": (c_int, [c_vp, c_size, c_int, c_int, ctypes.POINTER(c_vp)]),
    "mi355_allgather_f32": (c_int, [c_vp, c_vp, c_vp, c_size, c_vp]),
    "mi355_comm_destroy": (c_int, [c_vp]),
    "mi355_bicubic_rows_fwd": (c_int, [c_vp, c_vp] + [c_int] * 5 + [c_float, c_float, c_vp]),
    "mi355_stream_copy": (c_int, [c_vp, c_vp, c_size, c_vp]),
    "mi355_stream_read": (c_int, [c_vp, c_size, c_vp, c_vp]),
    "mi355_mfma_yardstick": (c_int, [c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355_event_time_begin": (c_int, [c_vp, ctypes.POINTER(c_vp)]),
    "mi355_event_time_end": (c_int, [c_vp, c_vp, ctypes.POINTER(c_float)]),
    "mi355_linear16_x32_fwd": (c_int, [c_vp] * 4 + [c_int] * 7 + [c_vp]),
    "mi355_linear16_stats_fwd": (c_int, [c_vp] * 5 + [c_int] * 6 + [c_vp, c_float, c_vp]),
    "mi355_linear16_ln16_fwd": (c_int, [c_vp] * 7 + [c_float, c_vp] + [c_int] * 7 + [c_vp]),
    "mi355_ln_lpi_stats_fwd": (c_int, [c_vp] * 10 + [c_float] + [c_vp] * 5 + [c_int] * 4 + [c_vp]),
    # SURVEY.md 8(b) spellings: aliases of mi355_sdpa_fwd / mi355_linear_fwd / mi355_mixer_token_fwd and the zero-byte workspace queries
    "mi355_sdpa_core_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, c_int, c_vp]),
    "mi355_sdpa_core_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_gemm_bias_act_fwd": (c_int, [c_vp] * 6 + [c_int] * 7 + [c_vp]),
    "mi355_gemm_bias_act_workspace_bytes": (c_size, [c_int] * 3),
    "mi355_mixer_token_mlp_fwd": (c_int, [c_vp] * 3 + [c_float] + [c_vp] * 5 + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "mi355_mixer_token_mlp_workspace_bytes": (c_size, [c_int] * 3),
    "mi355_cswin_lepe_attn_workspace_bytes": (c_size, [c_int] * 3),
    "mi355_xca_workspace_bytes": (c_size, [c_int] * 4),
    "mi355_layernorm_workspace_bytes": (c_size, [c_int] * 2),
}


class Mi355Error(RuntimeError):
    pass


class Mi355RangeError(Mi355Error):
    """A finite value saturated to inf in an fp16 operand tensor of an earlier launch (include/mi355attn.h mi355_range_status)."""


def lib():
    """Return the loaded library, loading (and type-annotating) it on first use.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(
                f"{LIB_PATH} is missing: build it with `python pytorch-attention_amd/build.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/eager fallback for the MI355X path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError here = header/library drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        v = handle.mi355_version()
        if v != ABI_VERSION:
            raise Mi355Error(f"libmi355attn ABI version {v} != binding version {ABI_VERSION}; rebuild")
        # the single-read SE / CBAM ops get dedicated workspaces from workspace_dedicated() below, so the promise holds -- on every
        # device this process uses (options are per device; the process default covers devices nobody has touched yet)
        handle.mi355_set_default_option(b"ws_persistent", 1)
        _lib = handle
    return _lib


MI355_EUNSUPPORTED = -2          # include/mi355attn.h: shape outside the kernel's envelope, nothing launched


def check(code, what):
    if code != 0:
        msg = lib().mi355_last_error()
        raise Mi355Error(f"{what} failed (code {code}): {msg.decode() if msg else '?'}")


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device`, as an int for ctypes.

    The library launches on the calling thread's CURRENT HIP device (include/mi355attn.h conventions), so a tensor that lives on
    another device than the current one is refused here instead of launching a kernel on the wrong GPU with foreign pointers."""
    if device is not None and device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
        raise Mi355Error(f"tensor on {device} but the current device is cuda:{torch.cuda.current_device()}: "
                         "wrap the call in `with torch.cuda.device(x.device):` (kernels launch on the current device)")
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_device_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise Mi355Error(
            f"{name} lives on {t.device}: the mi355attn modules only run on an MI355X device tensor "
            "(move the module and its input to 'cuda'); there is no CPU path in this package.")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.requires_grad and torch.is_grad_enabled():
        _warn_no_autograd()
    return t if t.is_contiguous() else t.contiguous()


_warned_autograd = False


def _warn_no_autograd():
    """Once per process: the engine is forward-only, its outputs carry no grad_fn."""
    global _warned_autograd
    if not _warned_autograd:
        _warned_autograd = True
        warnings.warn("mi355attn is a forward-only engine: a tensor that requires grad entered a kernel with autograd enabled; "
                      "the output has no grad_fn and no gradient will flow through this module (use torch.no_grad() / "
                      "requires_grad_(False) to silence this)", RuntimeWarning, stacklevel=4)


_ws_cache = {}


def _capturing():
    return torch.cuda.is_current_stream_capturing()


def workspace(nbytes, device):
    """Per-(device, stream) scratch tensor, grown on demand.  Reuse is safe because every op that uses it
    is enqueued on the same stream, in order.

    Under hipGraph capture the cache is bypassed: the buffer comes fresh from the graph's private memory pool, so a later eager
    call that grows or evicts a cached buffer can never free memory a captured graph still points at."""
    if _capturing():
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


_ws_named = {}


def workspace_named(name, nbytes, device):
    """Per-(name, device, stream) scratch tensor, grown on demand, kept apart from the generic workspace() buffer so that one
    large user (the GEMM's split-K slabs: tens of MB) does not inflate the scratch every small op borrows.  Capture-safe like
    workspace()."""
    if _capturing():
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (name, device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    t = _ws_named.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_named[key] = t
    return t


_ws_dedicated = collections.OrderedDict()
_WS_DEDICATED_MAX = int(os.environ.get("MI355_WS_CACHE", "64"))      # distinct (op, shape, stream) exchange workspaces kept alive


def workspace_dedicated(op_key, nbytes, device):
    """Workspace owned by ONE op + shape (single-read SE / CBAM): nothing else ever writes it, which is what lets the library
    skip re-zeroing the granule exchange area on every call ("ws_persistent", include/mi355attn.h).  A small LRU; an evicted
    buffer is reported to the library before it is released.  Under hipGraph capture: a fresh buffer from the graph's pool, never
    cached (see workspace())."""
    if _capturing():
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (op_key, device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    t = _ws_dedicated.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            lib().mi355_workspace_forget(ctypes.c_void_p(t.data_ptr()), t.numel())
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_dedicated[key] = t
        while len(_ws_dedicated) > _WS_DEDICATED_MAX:
            _, old = _ws_dedicated.popitem(last=False)
            lib().mi355_workspace_forget(ctypes.c_void_p(old.data_ptr()), old.numel())
    else:
        _ws_dedicated.move_to_end(key)
    return t


def set_option(key, value):
    check(lib().mi355_set_option(key.encode(), int(value)), f"set_option({key})")


def get_option(key):
    return lib().mi355_get_option(key.encode())


def kernel_trace(fn):
    """In-process kernel tally (include/mi355attn.h mi355_trace_begin / mi355_trace_end): run fn() with every instrumented launch
    timed by HIP events on its launch stream; returns [(tag, count, total_us, min_us, max_us)], largest total first."""
    check(lib().mi355_trace_begin(), "mi355_trace_begin")
    buf = ctypes.create_string_buffer(1 << 16)
    try:
        fn()
    finally:
        need = lib().mi355_trace_end(buf, len(buf))          # closes the trace also when fn() raised
    if need < 0:
        raise Mi355Error("mi355_trace_end failed (code %d)" % need)
    if need >= len(buf):                                      # did not fit: the library kept the report for a second call
        buf = ctypes.create_string_buffer(need + 1)
        lib().mi355_trace_end(buf, len(buf))
    rows = []
    for line in buf.value.decode().splitlines():
        parts = line.split("\t", 4)
        if len(parts) != 5:
            continue
        cnt, tot, mn, mx, tag = parts
        rows.append((tag, int(cnt), float(tot), float(mn), float(mx)))
    return rows


class StreamTimer:
    """HIP-event stopwatch on torch's current stream (events are recorded by the library on that stream)."""

    def __init__(self, device=None):
        self.device = device
        self.h = ctypes.c_void_p()

    def start(self):
        check(lib().mi355_event_time_begin(stream_ptr(self.device), ctypes.byref(self.h)), "event_time_begin")

    def stop_ms(self):
        ms = ctypes.c_float()
        check(lib().mi355_event_time_end(stream_ptr(self.device), self.h, ctypes.byref(ms)), "event_time_end")
        return float(ms.value)
