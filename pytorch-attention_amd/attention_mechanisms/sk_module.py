"""Drop-in for the reference's attention_mechanisms/sk_module.py: same import path, MI355X forward."""
from mi355attn.modules.axis import SKLayer  # noqa: F401
