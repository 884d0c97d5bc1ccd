"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a006_PaddingOperation import MyPadding` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import MyPadding  # noqa: F401
