from .scpoli import scpoli
from .scpoli_model import scPoli

__all__ = ["scpoli", "scPoli"]
