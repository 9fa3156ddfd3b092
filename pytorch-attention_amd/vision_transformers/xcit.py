"""Drop-in for the reference's vision_transformers/xcit.py: same import path, MI355X forward."""
from mi355attn.modules.xcit import (LPI, XCA, ClassAttention, ClassAttentionBlock, ConvPatchEmbed, Mlp,  # noqa: F401
                                    PositionalEncodingFourier, XCABlock, XCiT, conv3x3, xcit_nano_12_p16)
