"""Drop-in for the reference's mlps/mlp_mixer.py (layer level): same import path, MI355X forward."""
from mi355attn.modules.mixer import MLP_Mixer, MixerLayer, Mlp, PatchEmbedding  # noqa: F401
