from .trvae.unsupervised import trVAETrainer
from .expimap.regularized import expiMapTrainer
from .scpoli.trainer import scPoliTrainer

__all__ = ["trVAETrainer", "expiMapTrainer", "scPoliTrainer"]
