"""Split / dataset construction of the scPoli trainer (semantics of scarches/trainers/scpoli/_utils.py:103-282)."""
import numpy as np
import torch

from ...dataset import MultiConditionAnnotatedDataset
from ..trvae._utils import print_progress  # noqa: F401


def train_test_split(adata, train_frac=0.85, condition_keys=None, cell_type_key=None, labeled_array=None):
    """Train / validation indices.  With a cell-type key: per cell type of the labeled cells a shuffled split, unlabeled
    cells head / tail; with condition keys: per combined condition a SHUFFLED split (np.random, unseeded in the
    reference, `_utils.py:160`); otherwise one shuffled split."""
    indices = np.arange(adata.shape[0])
    if train_frac == 1:
        return indices, None

    def first_seen(values):
        seen = []
        for v in values:
            if v not in seen:
                seen.append(v)
        return seen

    tr, va = [], []
    if cell_type_key is not None:
        labeled = np.zeros(len(adata)) if labeled_array is None else np.ravel(labeled_array)
        lab_idx, unl_idx = indices[labeled == 1], indices[labeled == 0]
        if len(lab_idx) > 0:
            cts = np.asarray(adata.obs[cell_type_key])[lab_idx]
            for ct in first_seen(cts.tolist()):
                sel = lab_idx[cts == ct]
                k = int(np.ceil(train_frac * len(sel)))
                np.random.shuffle(sel)
                tr.append(sel[:k])
                va.append(sel[k:])
        if len(unl_idx) > 0:
            k = int(np.ceil(train_frac * len(unl_idx)))
            tr.append(unl_idx[:k])
            va.append(unl_idx[k:])
    elif condition_keys is not None:
        comb = np.asarray(adata.obs['conditions_combined'])
        for c in first_seen(comb.tolist()):
            sel = indices[comb == c]
            k = int(np.ceil(train_frac * len(sel)))
            np.random.shuffle(sel)
            tr.append(sel[:k])
            va.append(sel[k:])
    else:
        k = int(np.ceil(train_frac * len(indices)))
        np.random.shuffle(indices)
        return indices[:k], indices[k:]
    return np.concatenate(tr), np.concatenate(va)


def make_dataset(adata, train_frac=0.9, condition_keys=None, cell_type_keys=None, condition_encoders=None,
                 conditions_combined_encoder=None, cell_type_encoder=None, labeled_indices=None):
    """Split `adata` and wrap both parts as MultiConditionAnnotatedDataset (valid part None when train_frac == 1)."""
    labeled_array = np.zeros((len(adata), 1))
    if labeled_indices is not None:
        labeled_array[labeled_indices] = 1
    if cell_type_keys is not None:
        finest, n_max = None, 0
        for key in cell_type_keys:
            n = len(adata.obs[key].unique().tolist())
            if n >= n_max:
                n_max, finest = n, key
        train_idx, val_idx = train_test_split(adata, train_frac, cell_type_key=finest, labeled_array=labeled_array)
    elif condition_keys is not None:
        train_idx, val_idx = train_test_split(adata, train_frac, condition_keys=condition_keys)
    else:
        train_idx, val_idx = train_test_split(adata, train_frac)
    kw = dict(condition_keys=condition_keys, cell_type_keys=cell_type_keys, condition_encoders=condition_encoders,
              conditions_combined_encoder=conditions_combined_encoder, cell_type_encoder=cell_type_encoder)
    train = MultiConditionAnnotatedDataset(adata if train_frac == 1 else adata[train_idx],
                                           labeled_array=labeled_array[train_idx], **kw)
    if train_frac == 1:
        return train, None
    return train, MultiConditionAnnotatedDataset(adata[val_idx], labeled_array=labeled_array[val_idx], **kw)


def cov(x: torch.Tensor):
    """Unbiased covariance of the rows of x (numpy.cov(x, rowvar=False) semantics), for host-side callers."""
    if x.dim() == 1:
        x = x.view(-1, 1)
    xm = x - x.mean(0)
    return (xm.t() @ xm / (x.shape[0] - 1)).squeeze()
