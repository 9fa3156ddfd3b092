"""oracle/ -- CPU restatement of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and there only as the *checker*.  Nothing under ``pytorch-attention_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

What it is: an independent, functional (weights-in, tensor-out) restatement of the forward math of the
reference modules named by SURVEY.md section 8(a), written from the formulas in SURVEY.md section 9 with
explicit index math where the reference uses view/permute chains.  Every function cites the reference
file:line it follows (paths relative to the reference checkout).  It runs on torch-CPU tensors in fp32 or
fp64 (``dtype=`` argument) -- the reference's own arithmetic lives in ATen CPU kernels, so torch-CPU is
the faithful restatement vehicle for a floating-point path.

Pinning: the reference has no tests and pins no numeric result ("parity unpinned by
the reference's tests").  We pin the oracle ourselves against outputs of the *real* reference imported in
the build container under the seed protocol of SURVEY.md section 8(c): ``tests/golden/make_golden.py`` is
the generating script, ``tests/golden/golden.json`` + ``tests/golden/small/*.npz`` are its committed
outputs, and ``tests/test_oracle_golden.py`` checks every oracle function against them (``-m "not gpu"``).
"""
from .chan_attn import se_ex_forward  # noqa: F401
from .axis_attn import gc_forward, coordatt_forward, triplet_forward, bam_forward, sk_forward, pam_forward, cam_forward  # noqa: F401
from .chan_attn import simam_forward, srm_forward, gct_gauss_forward, lct_forward, gct_forward  # noqa: F401
from .chan_attn import se_forward, eca_forward, eca_kernel_size, cbam_forward, cbam_channel_forward, \
    cbam_spatial_forward, double_attention_forward, eca_gate_explicit, spatial_conv_explicit
from .transformer import (layernorm, gelu, linear, vit_attention_forward, vit_mlp_forward,
                          vit_encoder_forward, vit_patch_embed_forward, vit_forward, bicubic_rows, vit_position_rows,
                          mixer_layer_forward, sdpa_core, mixer_forward, mhsa_forward, global_attention_forward,
                          broad_attention_forward, qk_v_attention_forward, knn_attention_forward, conv_attention_forward,
                          pooling_attention_forward)
from .cswin import (lepe_attention_forward, cswin_block_forward, window_token_index, cswin_forward,  # noqa: F401
                    cswin_block_forward_aten, lepe_attention_forward_aten)
from .xcit import (xca_forward, lpi_forward, xca_block_forward, xcit_forward, conv_patch_embed_forward, fourier_position_rows,
                   class_attention_block_forward)
from .params import seeded_module_inputs, strip_prefix

__all__ = [n for n in dir() if not n.startswith("_")]
