from .expimap import expiMap
from .expimap_model import EXPIMAP

__all__ = ["expiMap", "EXPIMAP"]
