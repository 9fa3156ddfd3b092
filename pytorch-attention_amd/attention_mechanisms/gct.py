"""Drop-in for the reference's attention_mechanisms/gct.py: same import path, MI355X forward."""
from mi355attn.modules.zoo import GaussianGCT as GCT  # noqa: F401
