from .anndata import MultiConditionAnnotatedDataset

__all__ = ["MultiConditionAnnotatedDataset"]
