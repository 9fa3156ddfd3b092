"""Drop-in for the reference's attention_mechanisms/srm.py: same import path, MI355X forward."""
from mi355attn.modules.zoo import SRM  # noqa: F401
