"""Train/validation split and dataset construction with the reference's semantics
(scarches/trainers/trvae/_utils.py:73-205)."""
import sys

import numpy as np

from ...dataset import trVAEDataset


def print_progress(epoch, logs, n_epochs=10000, only_val_losses=True):
    """One-line progress bar with the current losses (keys containing 'unweighted' are not shown)."""
    parts = []
    for key, values in logs.items():
        if "unweighted" in key:
            continue
        if only_val_losses and "val_" not in key:
            continue
        parts.append(f" - {key:s}: {values[-1]:7.10f}")
    done, total, width = epoch + 1, n_epochs, 20
    filled = int(width * done // total)
    bar = "█" * filled + "-" * (width - filled)
    sys.stdout.write("\r |%s| %.1f%% %s" % (bar, 100 * (done / float(total)), "".join(parts)))
    if done == total:
        sys.stdout.write("\n")
    sys.stdout.flush()


def train_test_split(adata, train_frac=0.85, condition_key=None, cell_type_key=None, labeled_array=None):
    """Indices of the training and validation cells.

    * `condition_key`: per condition (in order of first appearance) the FIRST ceil(train_frac * n_c) cells train and
      the tail validates - no shuffling, exactly like the reference (`_utils.py:123-135`).
    * `cell_type_key`: per cell type of the labeled cells a shuffled split, unlabeled cells head/tail.
    * neither: one shuffled split.
    """
    indices = np.arange(adata.shape[0])
    if train_frac == 1:
        return indices, None
    if cell_type_key is not None:
        labeled = np.zeros(len(adata)) if labeled_array is None else np.ravel(labeled_array)
        lab_idx, unl_idx = indices[labeled == 1], indices[labeled == 0]
        tr, va = [], []
        if len(lab_idx) > 0:
            cts = np.asarray(adata.obs[cell_type_key])[lab_idx]
            seen = []
            for ct in cts.tolist():
                if ct not in seen:
                    seen.append(ct)
            for ct in seen:
                sel = lab_idx[cts == ct]
                k = int(np.ceil(train_frac * len(sel)))
                np.random.shuffle(sel)
                tr.append(sel[:k])
                va.append(sel[k:])
        if len(unl_idx) > 0:
            k = int(np.ceil(train_frac * len(unl_idx)))
            tr.append(unl_idx[:k])
            va.append(unl_idx[k:])
        return np.concatenate(tr), np.concatenate(va)
    if condition_key is not None:
        cond = np.asarray(adata.obs[condition_key])
        order = []
        for c in cond.tolist():
            if c not in order:
                order.append(c)
        tr, va = [], []
        for c in order:
            sel = indices[cond == c]
            k = int(np.ceil(train_frac * len(sel)))
            tr.append(sel[:k])
            va.append(sel[k:])
        return np.concatenate(tr), np.concatenate(va)
    k = int(np.ceil(train_frac * len(indices)))
    np.random.shuffle(indices)
    return indices[:k], indices[k:]


def make_dataset(adata, train_frac=0.9, condition_key=None, cell_type_keys=None, condition_encoder=None,
                 cell_type_encoder=None, labeled_indices=None):
    """Split `adata` and wrap both parts as trVAEDataset objects (valid part is None when train_frac == 1)."""
    print(f"Preparing {adata.shape}")
    labeled_array = np.zeros((len(adata), 1))
    if labeled_indices is not None:
        labeled_array[labeled_indices] = 1
    if cell_type_keys is not None:
        finest, n_max = None, 0
        for key in cell_type_keys:
            n = len(adata.obs[key].unique().tolist())
            if n >= n_max:
                n_max, finest = n, key
        print(f"Splitting data {adata.shape}")
        train_idx, val_idx = train_test_split(adata, train_frac, cell_type_key=finest, labeled_array=labeled_array)
    elif condition_key is not None:
        train_idx, val_idx = train_test_split(adata, train_frac, condition_key=condition_key)
    else:
        train_idx, val_idx = train_test_split(adata, train_frac)

    print("Instantiating dataset")
    kw = dict(condition_key=condition_key, cell_type_keys=cell_type_keys, condition_encoder=condition_encoder,
              cell_type_encoder=cell_type_encoder)
    train = trVAEDataset(adata if train_frac == 1 else adata[train_idx], labeled_array=labeled_array[train_idx], **kw)
    if train_frac == 1:
        return train, None
    valid = trVAEDataset(adata[val_idx], labeled_array=labeled_array[val_idx], **kw)
    return train, valid
