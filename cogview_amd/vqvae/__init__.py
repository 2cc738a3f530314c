from .api import code2img, img2code, new_model  # noqa: F401
from .vqvae_zc import VQVAE  # noqa: F401
