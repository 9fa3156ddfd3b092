"""Drop-in for the attention module of the reference's vision_transformers/p2t.py (module level only -- the rest of
that file is not mirrored): same import path and class name, MI355X forward."""
from mi355attn.modules.mhsa import PoolingAttention  # noqa: F401
