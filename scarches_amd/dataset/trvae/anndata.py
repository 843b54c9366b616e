"""Dataset of the trVAE trainer: the CSR minibatch loader.

API of scarches/dataset/trvae/anndata.py:12-123 (`AnnotatedDataset`): same constructor, `__getitem__` dict
{x, labeled, sizefactor, batch[, celltypes]}, `__len__`, `stratifier_weights`.  The difference is where batches are
built: instead of B scipy row slices + toarray() per step on the host, the whole matrix is uploaded ONCE as CSR
(`to_device`) and minibatches are assembled in HBM by the `sca_csr_gather_rows` kernel from a vector of row ids.
"""
from collections import Counter

import numpy as np
import scipy.sparse as sp
import torch

from ._utils import label_encoder


class AnnotatedDataset(torch.utils.data.Dataset):
    def __init__(self, adata, condition_key=None, condition_encoder=None, cell_type_keys=None,
                 cell_type_encoder=None, labeled_array=None):
        self.condition_key = condition_key
        self.condition_encoder = condition_encoder
        self.cell_type_keys = cell_type_keys
        self.cell_type_encoder = cell_type_encoder

        X = adata.X
        self._is_sparse = sp.issparse(X)
        # canonical CSR, float32 values / int32 columns / int64 row pointers: the layout the kernels read
        csr = X.tocsr().astype(np.float32) if self._is_sparse else sp.csr_matrix(np.asarray(X, dtype=np.float32))
        if not csr.has_canonical_format:
            csr = csr.copy()
            csr.sum_duplicates()
        self.csr = csr
        self.data = X if self._is_sparse else torch.tensor(np.asarray(X))
        self.size_factors = torch.tensor(np.ravel(X.sum(1)))

        labeled_array = np.zeros((len(adata), 1)) if labeled_array is None else labeled_array
        self.labeled_vector = torch.tensor(labeled_array)

        if self.condition_key is not None:
            cond = label_encoder(adata, encoder=self.condition_encoder, condition_key=condition_key)
            self.conditions = torch.tensor(cond, dtype=torch.long)
        if self.cell_type_keys is not None:
            levels = [label_encoder(adata, encoder=self.cell_type_encoder, condition_key=k) for k in cell_type_keys]
            self.cell_types = torch.tensor(np.stack(levels).T, dtype=torch.long)
        self._dev = None

    # ---- reference-compatible host access ---------------------------------------------------------------
    def __getitem__(self, index):
        out = dict()
        if self._is_sparse:
            out["x"] = torch.tensor(np.squeeze(self.data[index].toarray()))
        else:
            out["x"] = self.data[index]
        out["labeled"] = self.labeled_vector[index]
        out["sizefactor"] = self.size_factors[index]
        if self.condition_key:
            out["batch"] = self.conditions[index]
        if self.cell_type_keys:
            out["celltypes"] = self.cell_types[index, :]
        return out

    def __len__(self):
        return self.csr.shape[0]

    @property
    def condition_label_encoder(self) -> dict:
        return self.condition_encoder

    @condition_label_encoder.setter
    def condition_label_encoder(self, value: dict):
        if value is not None:
            self.condition_encoder = value

    @property
    def cell_type_label_encoder(self) -> dict:
        return self.cell_type_encoder

    @cell_type_label_encoder.setter
    def cell_type_label_encoder(self, value: dict):
        if value is not None:
            self.cell_type_encoder = value

    @property
    def stratifier_weights(self):
        """w_i = 1 / (N * count(condition_i)), float64 (anndata.py:115-123)."""
        cond = self.conditions.detach().cpu().numpy()
        count = Counter(cond.tolist())
        per_cell = np.array([count[c] for c in cond.tolist()])
        return ((1.0 / len(cond)) / per_cell).astype(float)

    # ---- device residency ---------------------------------------------------------------------------------
    def to_device(self, device):
        """Upload the CSR triplet and the condition vector once; returns a dict of device tensors."""
        if self._dev is not None and self._dev["device"] == torch.device(device):
            return self._dev
        csr = self.csr
        d = dict(device=torch.device(device),
                 indptr=torch.from_numpy(csr.indptr.astype(np.int64)).to(device),
                 indices=torch.from_numpy(csr.indices.astype(np.int32)).to(device),
                 data=torch.from_numpy(csr.data.astype(np.float32)).to(device))
        if self.condition_key is not None:
            d["cond"] = self.conditions.to(torch.int32).to(device)
        else:
            d["cond"] = torch.zeros(len(self), dtype=torch.int32, device=device)
        if self.cell_type_keys is not None:
            d["celltypes"] = self.cell_types.to(device)
        self._dev = d
        return d
