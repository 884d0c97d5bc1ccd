"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a003_AutoPathMLP import AutoPathMLP` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import AutoPathMLP  # noqa: F401
