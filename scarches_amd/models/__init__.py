from .trvae.trvae import trVAE
from .trvae.trvae_model import TRVAE
from .trvae.adaptors import Adaptor, attach_adaptors
from .expimap.expimap import expiMap
from .expimap.expimap_model import EXPIMAP
from .scpoli.scpoli import scpoli
from .scpoli.scpoli_model import scPoli

__all__ = ["trVAE", "TRVAE", "expiMap", "EXPIMAP", "scpoli", "scPoli", "Adaptor", "attach_adaptors"]
