from .chan_attn import CBAM, ChannelAttention, DoubleAttention, ECALayer, SELayer, SpatialAttention  # noqa: F401
from .cswin import CSWinBlock, LePEAttention  # noqa: F401
from .mixer import MixerLayer  # noqa: F401
from .vit import Attention, PatchEmbedding, TransformerEncoder, VisionTransformer  # noqa: F401
from .xcit import LPI, XCA, XCABlock  # noqa: F401
