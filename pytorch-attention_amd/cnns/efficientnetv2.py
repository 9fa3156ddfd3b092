"""Drop-in for the squeeze-excite module of the reference's cnns/efficientnetv2.py:14-29 (module level only -- the network
around it is not mirrored): same import path and class name, MI355X forward."""
from mi355attn.modules.se_variants import SELayerBias4 as SELayer  # noqa: F401
