"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a010_StateRecorder import StateRecorder` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import StateRecorder  # noqa: F401
