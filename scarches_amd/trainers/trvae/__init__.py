from .trainer import Trainer
from .unsupervised import trVAETrainer

__all__ = ["Trainer", "trVAETrainer"]
