"""Drop-in for the reference's attention_mechanisms/gate_channel_module.py: same import path, MI355X forward."""
from mi355attn.modules.zoo import GCT  # noqa: F401
