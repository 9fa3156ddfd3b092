"""Drop-in for the reference's attention_mechanisms/dual_attention.py: same import path, MI355X forward."""
from mi355attn.modules.axis import CAM, PAM  # noqa: F401
