"""Oracle (test infrastructure): the seed protocol of SURVEY.md section 8(c), defined once.

    torch.manual_seed(1234); m = Ctor(...).eval(); torch.manual_seed(4321); x = torch.randn(*shape)

The CPU generator is deterministic for a given torch build, so the GPU box regenerates bit-identical
weights and inputs from the two seeds; no reference checkout is needed at test time.
"""
import torch

WEIGHT_SEED = 1234
INPUT_SEED = 4321


def seeded_module_inputs(ctor, shape):
    """Build ``ctor()`` under WEIGHT_SEED (eval mode) and a randn input of ``shape`` under INPUT_SEED."""
    torch.manual_seed(WEIGHT_SEED)
    module = ctor().eval()
    torch.manual_seed(INPUT_SEED)
    x = torch.randn(*shape)
    return module, x


def strip_prefix(state, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in state.items() if k.startswith(prefix)}
