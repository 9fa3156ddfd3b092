"""Drop-in for the reference's attention_mechanisms/se_module.py: same import path, MI355X forward."""
from mi355attn.modules.chan_attn import SELayer  # noqa: F401
