"""Drop-in shim: put this directory on sys.path and the reference's import line
`from a011_PatchOperation import PatchMergingAndLinearLayer` resolves to the HIP-backed implementation."""
from swin_unet_image_fusion_amd.modules import PatchMergingAndLinearLayer  # noqa: F401
