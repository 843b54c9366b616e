from .trvae.anndata import AnnotatedDataset as trVAEDataset
from .scpoli.anndata import MultiConditionAnnotatedDataset

__all__ = ["trVAEDataset", "MultiConditionAnnotatedDataset"]
