from .trvae import trVAE
from .trvae_model import TRVAE
from .adaptors import Adaptor, attach_adaptors

__all__ = ["trVAE", "TRVAE", "Adaptor", "attach_adaptors"]
