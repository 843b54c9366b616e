from .trainer import scPoliTrainer

__all__ = ["scPoliTrainer"]
