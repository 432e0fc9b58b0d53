from .autoencoder import AutoencodingEngine
from .diffusion import DiffusionEngine
