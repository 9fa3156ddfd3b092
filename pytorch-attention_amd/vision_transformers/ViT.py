"""Drop-in for the reference's vision_transformers/ViT.py: same import path, MI355X forward."""
from mi355attn.modules.vit import Attention, Mlp, PatchEmbedding, TransformerEncoder, VisionTransformer  # noqa: F401
