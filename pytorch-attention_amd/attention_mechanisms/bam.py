"""Drop-in for the reference's attention_mechanisms/bam.py: same import path, MI355X forward."""
from mi355attn.modules.axis import ChannelGate, SpatialGate, BAM  # noqa: F401
