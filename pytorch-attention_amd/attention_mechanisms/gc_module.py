"""Drop-in for the reference's attention_mechanisms/gc_module.py: same import path, MI355X forward."""
from mi355attn.modules.axis import GCModule  # noqa: F401
