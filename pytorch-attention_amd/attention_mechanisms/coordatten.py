"""Drop-in for the reference's attention_mechanisms/coordatten.py: same import path, MI355X forward."""
from mi355attn.modules.axis import CoordinateAttention  # noqa: F401
