"""Drop-in for the reference's vision_transformers/cswin.py (block level): same import path, MI355X forward."""
from mi355attn.modules.cswin import CSWinBlock, LePEAttention, Mlp  # noqa: F401
