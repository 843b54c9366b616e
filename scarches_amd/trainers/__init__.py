from .trvae.unsupervised import trVAETrainer
from .expimap.regularized import expiMapTrainer

__all__ = ["trVAETrainer", "expiMapTrainer"]
