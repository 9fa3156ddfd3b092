"""Drop-in for the reference's vision_transformers/cswin.py (block level): same import path, MI355X forward."""
from mi355attn.modules.cswin import (CSWin_64_12211_tiny_224, CSWin_64_24322_small_224, CSWinBlock, CSWinTransformer,  # noqa: F401
                                     LePEAttention, Merge_Block, Mlp)
