"""Drop-in for the attention module of the reference's vision_transformers/moat.py (module level only -- the rest of
that file is not mirrored): same import path and class name, MI355X forward."""
from mi355attn.modules.chan_attn import SELayer  # noqa: F401  (moat.py:18-33 is the plain bias-free SELayer)
from mi355attn.modules.mhsa import Attention  # noqa: F401
