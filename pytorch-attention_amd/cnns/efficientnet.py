"""Drop-in for the squeeze-excite module of the reference's cnns/efficientnet.py (module level only -- the network around it is
not mirrored): same import path and class name, MI355X forward."""
from mi355attn.modules.se_variants import SELayerBias as SELayer  # noqa: F401
