"""MI355X-native forward path of the Swin-UNet IR/visible image-fusion network.

Public surface = the reference's module API (a013 `MyModel`, a012 `SelfAndCrossBlockPair`,
a001 `WindowAttention` and the inner modules needed for state_dict compatibility), implemented
by hand-written HIP kernels behind a C-ABI (`include/swinfuse.h`, `libswinfuse.so`).
"""
from .config import CONFIGS, FusionConfig, load_recipe_into, synthetic_pair  # noqa: F401
from .modules import (AddAndLayerNormWithOtherModule, AutoPathMLP, AutoPathWinAtt, BasicBlock, MyModel,  # noqa: F401
                      MyPadding, NormalAndShiftWinsBlockPair, PatchMergingAndLinearLayer, SelfAndCrossBlockPair,
                      StateRecorder, WindowAttention, get_encoder_or_decoder_block)

__version__ = "0.1.0"
