"""Drop-in for the reference's vision_transformers/xcit.py (block level): same import path, MI355X forward."""
from mi355attn.modules.xcit import LPI, XCA, Mlp, XCABlock  # noqa: F401
