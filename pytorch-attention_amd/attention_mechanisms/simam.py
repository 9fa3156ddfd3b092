"""Drop-in for the reference's attention_mechanisms/simam.py: same import path, MI355X forward."""
from mi355attn.modules.zoo import simam_module  # noqa: F401
