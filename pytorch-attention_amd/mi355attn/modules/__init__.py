from .chan_attn import CBAM, ChannelAttention, DoubleAttention, ECALayer, SELayer, SpatialAttention  # noqa: F401
from .cswin import (CSWin_64_12211_tiny_224, CSWin_64_24322_small_224, CSWinBlock, CSWinTransformer,  # noqa: F401
                    LePEAttention, Merge_Block)
from .mixer import MLP_Mixer, MixerLayer  # noqa: F401
from .vit import Attention, PatchEmbedding, TransformerEncoder, VisionTransformer  # noqa: F401
from .xcit import (LPI, XCA, ClassAttention, ClassAttentionBlock, ConvPatchEmbed, PositionalEncodingFourier, XCABlock, XCiT,  # noqa: F401
                   xcit_nano_12_p16)
from .zoo import GCT, LCT, SRM, GaussianGCT, simam_module  # noqa: F401
from .mhsa import (Broad_Attention, ConvAttention, GlobalAttention, KNNAttention, PoolingAttention, QKVSplitAttention, SRAttention,  # noqa: F401
                   SRAttentionRelPos, SRConvAttention)
from .se_variants import SELayerBias, SELayerBias4, SELayerHidden, SqueezeExcite  # noqa: F401
from .axis import BAM, CAM, PAM, CoordinateAttention, GCModule, SKLayer, TripletAttention  # noqa: F401


def _guard_all():
    """Every drop-in class defined in this package gets the reference's behaviour on large activations: forward() falls back to strict
    mode when fp16 operands saturate (functional.range_fallback_forward; option "range_fallback")."""
    import sys
    from torch import nn
    from .. import functional as F
    for modname, mod in list(sys.modules.items()):
        if modname.startswith(__name__ + ".") and mod is not None:
            for obj in list(vars(mod).values()):
                if isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__ == modname:
                    F.range_guarded(obj)


_guard_all()
