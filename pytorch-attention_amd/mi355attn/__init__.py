"""mi355attn -- MI355X-native (gfx950) forward engine behind the pytorch-attention nn.Module surface.

Host side of the drop-in: ``mi355attn.modules`` mirrors the reference's classes (same names, ctor
signatures, state_dict layouts, forward signatures); ``mi355attn.functional`` wraps the C ABI of
``libmi355attn.so`` (include/mi355attn.h).  No CPU fallback exists in this package.
"""
from . import functional  # noqa: F401
from ._ffi import LIB_PATH, Mi355Error, Mi355RangeError, StreamTimer, get_option, kernel_trace, lib, set_option  # noqa: F401
from .functional import (PREC_BF16, PREC_FP16, PREC_STRICT, default_precision, guarded_forward, range_status,  # noqa: F401
                         set_default_precision, sync_status)

__version__ = "0.1.0"
