from .monitor import EarlyStopping
from .knn import weighted_knn_trainer, weighted_knn_transfer

__all__ = ["EarlyStopping", "weighted_knn_trainer", "weighted_knn_transfer"]
