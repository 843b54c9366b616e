from .trvae.unsupervised import trVAETrainer

__all__ = ["trVAETrainer"]
