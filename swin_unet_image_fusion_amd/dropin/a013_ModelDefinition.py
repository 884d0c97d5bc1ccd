"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a013_ModelDefinition import MyModel` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import MyModel, get_encoder_or_decoder_block  # noqa: F401
