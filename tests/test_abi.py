"""CPU tests: the C-ABI library builds, loads and exports exactly what include/mi355attn.h declares."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mi355attn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", src)) - {"mi355_stream_t"})


def _declared_arity():
    """function name -> number of parameters, parsed from the header's prototypes."""
    src = open(os.path.join(ROOT, "include", "mi355attn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(mi355_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_binding_arity_matches_header():
    """ctypes cannot check a prototype: a wrong argtypes length only fails at call time on the GPU box -- catch it here."""
    import mi355attn._ffi as ffi
    arity = _declared_arity()
    assert sorted(arity) == sorted(ffi.SIGNATURES)
    bad = {n: (len(a), arity[n]) for n, (_, a) in ffi.SIGNATURES.items() if len(a) != arity[n]}
    assert not bad, f"argtypes length != header parameter count: {bad}"


def test_header_symbols_exported(built_lib):
    handle = ctypes.CDLL(built_lib)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in mi355attn.h but not exported: {missing}"


def test_binding_table_matches_header(built_lib):
    import mi355attn._ffi as ffi
    assert sorted(ffi.SIGNATURES) == _declared()
    lib = ffi.lib()
    assert lib.mi355_version() == ffi.ABI_VERSION


def test_error_text_and_options(built_lib):
    import mi355attn
    from mi355attn import _ffi
    lib = _ffi.lib()
    assert lib.mi355_set_option(b"no_such_key", 1) == -1
    assert b"no_such_key" in lib.mi355_last_error()
    old = mi355attn.get_option("chunk_images")
    mi355attn.set_option("chunk_images", 7)
    assert mi355attn.get_option("chunk_images") == 7
    mi355attn.set_option("chunk_images", old)
    # argument validation happens before any HIP call, so it is testable without a GPU
    rc = lib.mi355_se_fwd(None, None, None, None, 1, 1, 1, 1, 1, None, 0, None)
    assert rc == -1 and b"invalid argument" in lib.mi355_last_error()


def test_no_cpu_fallback(built_lib):
    """A CPU tensor must raise, never silently compute somewhere else."""
    import pytest
    import torch
    from mi355attn import Mi355Error
    from mi355attn.modules import SELayer
    with pytest.raises(Mi355Error):
        SELayer(64)(torch.randn(2, 64, 8, 8))


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files import the oracle: {bad}"


def test_options_are_per_device_with_process_defaults(built_lib):
    """mi355_set_option acts on the calling thread's current device, mi355_set_default_option on what devices without an own setting
    read (no process-global knobs).  Without a GPU the current device is ordinal 0, which is enough to
    check the override / default semantics, the range checks and that the tuning-only GEMM variants are no longer reachable."""
    import ctypes
    lib = ctypes.CDLL(built_lib)
    lib.mi355_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    lib.mi355_set_default_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    lib.mi355_get_option.argtypes = [ctypes.c_char_p]
    lib.mi355_get_option.restype = ctypes.c_long
    lib.mi355_last_error.restype = ctypes.c_char_p
    # (the library may already be loaded in this process by the binding, which sets the DEFAULT of "ws_persistent": use another key)
    assert lib.mi355_get_option(b"nt") == 3 and lib.mi355_get_option(b"reverse") == 0 and lib.mi355_get_option(b"ln_fold") == 0
    assert lib.mi355_set_default_option(b"reverse", 1) == 0                # what the Python binding does for "ws_persistent" at load time
    assert lib.mi355_get_option(b"reverse") == 1                           # no override on this device: the default shows
    assert lib.mi355_set_option(b"reverse", 0) == 0                        # override on the current device wins ...
    assert lib.mi355_get_option(b"reverse") == 0
    assert lib.mi355_set_default_option(b"reverse", 1) == 0                # ... also over a later change of the default
    assert lib.mi355_get_option(b"reverse") == 0
    assert lib.mi355_set_default_option(b"reverse", 0) == 0
    assert lib.mi355_set_option(b"gemm_variant", 17) == 0 and lib.mi355_get_option(b"gemm_variant") == 17   # 17: gemm16_w4.hip (round 5)
    for bad in (18, 21, 31, -1):                                           # timing ablations that produce wrong results never ship as option values
        assert lib.mi355_set_option(b"gemm_variant", bad) == -1
        assert b"gemm_variant" in lib.mi355_last_error()
    assert lib.mi355_get_option(b"gemm_variant") == 17
    assert lib.mi355_set_option(b"gemm_variant", 0) == 0
    assert lib.mi355_set_option(b"spin_limit", 0) == 0 and lib.mi355_set_option(b"spin_limit", 5) == -1
    assert lib.mi355_set_option(b"spin_limit", 1 << 22) == 0
    assert lib.mi355_set_option(b"no_such_key", 1) == -1 and lib.mi355_get_option(b"no_such_key") == -1


def test_c2_kernels_have_no_scratch(built_lib):
    """the single-read SE / ECA / CBAM kernels of the C2 bench shape (56 x 56: 13 float4 per lane, full bands)
    must not spill -- read from the AMDGPU metadata of the shipped library (tools/kernel_resources.py), no GPU needed."""
    import sys
    pytest.importorskip("msgpack")                                   # tools/kernel_resources.py decodes the metadata notes with it
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    rows = kernel_resources.kernels(built_lib)
    assert len(rows) > 300
    c2 = [r for r in rows if ("se_single_kernel<13," in r["demangled"] or "eca_halo_kernel<" in r["demangled"]
                              or ("cbam_single_kernel<" in r["demangled"] and r["demangled"].split("(")[0].rstrip(">").endswith("true")))]
    assert len(c2) >= 10, [r["demangled"] for r in c2][:5]
    bad = [(r["demangled"], r["scratch"]) for r in c2 if r["scratch"]]
    assert not bad, bad
    se3 = [r for r in rows if "se_single_kernel<13, true, true, 3, false>" in r["demangled"]]        # the bench instantiation: three workgroups per CU
    assert se3 and se3[0]["vgpr"] <= 80 and se3[0]["scratch"] == 0, se3
