"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a005_BasicBlock import BasicBlock` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import BasicBlock  # noqa: F401
