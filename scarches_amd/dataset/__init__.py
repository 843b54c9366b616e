from .trvae.anndata import AnnotatedDataset as trVAEDataset

__all__ = ["trVAEDataset"]
