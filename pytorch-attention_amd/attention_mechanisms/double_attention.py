"""Drop-in for the reference's attention_mechanisms/double_attention.py: same import path, MI355X forward."""
from mi355attn.modules.chan_attn import DoubleAttention  # noqa: F401
