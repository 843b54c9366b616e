from .anndata import AnnotatedDataset
from ._utils import label_encoder

__all__ = ["AnnotatedDataset", "label_encoder"]
