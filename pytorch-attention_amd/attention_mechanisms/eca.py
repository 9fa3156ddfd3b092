"""Drop-in for the reference's attention_mechanisms/eca.py: same import path, MI355X forward."""
from mi355attn.modules.chan_attn import ECALayer  # noqa: F401
