from .autoencoder import AutoencoderKL, AutoencoderKLModeOnly  # noqa: F401
from .diffusion import DiffusionEngine  # noqa: F401
