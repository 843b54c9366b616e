from .monitor import EarlyStopping

__all__ = ["EarlyStopping"]
