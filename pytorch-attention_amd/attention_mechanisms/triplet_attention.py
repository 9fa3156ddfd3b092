"""Drop-in for the reference's attention_mechanisms/triplet_attention.py: same import path, MI355X forward."""
from mi355attn.modules.axis import BasicConv2d, ZPool, AttentionGate, TripletAttention  # noqa: F401
