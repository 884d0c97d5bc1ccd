"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a012_SelfAndCrossBlockPair import SelfAndCrossBlockPair` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import SelfAndCrossBlockPair  # noqa: F401
