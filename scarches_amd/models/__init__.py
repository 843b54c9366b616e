from .trvae.trvae import trVAE
from .trvae.trvae_model import TRVAE

__all__ = ["trVAE", "TRVAE"]
