"""Drop-in for the reference's attention_mechanisms/cbam.py: same import path, MI355X forward."""
from mi355attn.modules.chan_attn import CBAM, ChannelAttention, SpatialAttention  # noqa: F401
