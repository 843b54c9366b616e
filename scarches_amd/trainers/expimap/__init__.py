from .regularized import ProxGroupLasso, ProxL1, expiMapTrainer

__all__ = ["expiMapTrainer", "ProxGroupLasso", "ProxL1"]
