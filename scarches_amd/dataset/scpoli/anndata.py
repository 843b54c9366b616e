"""Dataset of the scPoli trainer: CSR minibatch loader with several condition covariates.

API of scarches/dataset/scpoli/anndata.py:10-129 (`MultiConditionAnnotatedDataset`): `x` float32, `batch`
[n_covariates] codes (one per condition key), `combined_batch` code of the joined condition, `celltypes` [n_levels],
`labeled`, `sizefactor`.  As for trVAE the matrix is uploaded once as CSR (`to_device`) and minibatches are assembled in
HBM from row ids.
"""
from collections import Counter

import numpy as np
import scipy.sparse as sp
import torch

from ..trvae._utils import label_encoder


class MultiConditionAnnotatedDataset(torch.utils.data.Dataset):
    def __init__(self, adata, condition_keys=None, condition_encoders=None, conditions_combined_encoder=None,
                 cell_type_keys=None, cell_type_encoder=None, labeled_array=None):
        self.condition_keys = condition_keys
        self.condition_encoders = condition_encoders
        self.conditions_combined_encoder = conditions_combined_encoder
        self.cell_type_keys = cell_type_keys
        self.cell_type_encoder = cell_type_encoder

        X = adata.X
        self._is_sparse = sp.issparse(X)
        csr = X.tocsr().astype(np.float32) if self._is_sparse else sp.csr_matrix(np.asarray(X, dtype=np.float32))
        if not csr.has_canonical_format:
            csr = csr.copy()
            csr.sum_duplicates()
        self.csr = csr
        self.data = X if self._is_sparse else torch.tensor(np.asarray(X), dtype=torch.float32)
        self.size_factors = torch.tensor(np.ravel(X.sum(1)))
        labeled_array = np.zeros((len(adata), 1)) if labeled_array is None else labeled_array
        self.labeled_vector = torch.tensor(labeled_array)

        if self.condition_keys is not None:
            codes = [label_encoder(adata, encoder=self.condition_encoders[k], condition_key=k)
                     for k in condition_keys[:len(self.condition_encoders)]]
            self.conditions = torch.tensor(np.stack(codes), dtype=torch.long).T
            comb = label_encoder(adata, encoder=self.conditions_combined_encoder, condition_key='conditions_combined')
            self.conditions_combined = torch.tensor(comb, dtype=torch.long)
        if self.cell_type_keys is not None:
            levels = [label_encoder(adata, encoder=self.cell_type_encoder, condition_key=k) for k in cell_type_keys]
            self.cell_types = torch.tensor(np.stack(levels).T, dtype=torch.long)
        self._dev = None

    def __getitem__(self, index):
        out = dict()
        if self._is_sparse:
            out["x"] = torch.tensor(np.squeeze(self.data[index].toarray()), dtype=torch.float32)
        else:
            out["x"] = self.data[index]
        out["labeled"] = self.labeled_vector[index]
        out["sizefactor"] = self.size_factors[index]
        if self.condition_keys:
            out["batch"] = self.conditions[index, :]
            out["combined_batch"] = self.conditions_combined[index]
        if self.cell_type_keys:
            out["celltypes"] = self.cell_types[index, :]
        return out

    def __len__(self):
        return self.csr.shape[0]

    @property
    def stratifier_weights(self):
        cond = self.conditions.detach().cpu().numpy().squeeze()
        keys = [tuple(np.atleast_1d(c).tolist()) for c in cond]
        count = Counter(keys)
        return ((1.0 / len(keys)) / np.array([count[k] for k in keys])).astype(float)

    def to_device(self, device):
        """Upload CSR + label vectors once; returns a dict of device tensors."""
        if self._dev is not None and self._dev["device"] == torch.device(device):
            return self._dev
        csr, n = self.csr, len(self)
        d = dict(device=torch.device(device),
                 indptr=torch.from_numpy(csr.indptr.astype(np.int64)).to(device),
                 indices=torch.from_numpy(csr.indices.astype(np.int32)).to(device),
                 data=torch.from_numpy(csr.data.astype(np.float32)).to(device),
                 labeled=(self.labeled_vector.reshape(-1) == 1).to(torch.uint8).to(device))
        if self.condition_keys is not None:
            d["covs"] = self.conditions.to(torch.int32).contiguous().to(device)
            d["cond"] = self.conditions_combined.to(torch.int32).to(device)
        else:
            d["covs"] = torch.zeros(n, 0, dtype=torch.int32, device=device)
            d["cond"] = torch.zeros(n, dtype=torch.int32, device=device)
        if self.cell_type_keys is not None:
            d["celltypes"] = self.cell_types.to(torch.int32).contiguous().to(device)
        self._dev = d
        return d
