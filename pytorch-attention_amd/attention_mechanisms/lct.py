"""Drop-in for the reference's attention_mechanisms/lct.py: same import path, MI355X forward."""
from mi355attn.modules.zoo import LCT  # noqa: F401
