from .trvae import trVAE
from .trvae_model import TRVAE

__all__ = ["trVAE", "TRVAE"]
