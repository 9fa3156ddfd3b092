"""CPU tests: the C-ABI library builds, loads and exports exactly what include/mi355attn.h declares."""
import ctypes
import os
import re

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
