"""A minimal in-memory stand-in for `anndata.AnnData` (anndata is an optional dependency of this package).

The model / trainer code only needs: `X` (dense ndarray or scipy CSR), `obs` (pandas DataFrame), `var_names`
(pandas Index), `n_vars`, `shape`, `len()`, row indexing and `[:, gene_names]` column selection.  A real AnnData
works everywhere a CellData does (duck typing); CellData exists so that the path has no hard anndata dependency.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import scipy.sparse as sp


class CellData:
    def __init__(self, X, obs: pd.DataFrame | None = None, var_names=None, dtype=None):
        if dtype is not None and not sp.issparse(X):
            X = np.asarray(X, dtype=dtype)
        elif dtype is not None:
            X = X.astype(dtype)
        self.X = X
        n, g = X.shape
        self.obs = obs.reset_index(drop=True) if obs is not None else pd.DataFrame(index=pd.RangeIndex(n))
        if len(self.obs) != n:
            raise ValueError("obs must have one row per cell")
        self.var_names = pd.Index(var_names if var_names is not None else [f"gene{i}" for i in range(g)]).astype(str)
        if len(self.var_names) != g:
            raise ValueError("var_names must have one entry per gene")
        self.varm, self.uns, self.obsm = {}, {}, {}

    @property
    def shape(self):
        return self.X.shape

    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, key):
        if isinstance(key, tuple):
            rows, cols = key
            sub = self if (isinstance(rows, slice) and rows == slice(None)) else self[rows]
            if isinstance(cols, slice):
                ci = np.arange(self.n_vars)[cols]
            else:
                cols = pd.Index(cols)
                ci = self.var_names.get_indexer(cols.astype(str)) if cols.dtype == object else np.asarray(cols)
                if (ci < 0).any():
                    raise KeyError("unknown gene name in column selection")
            X = sub.X[:, ci]
            out = CellData(X, sub.obs, self.var_names[ci])
            return out
        if isinstance(key, slice):
            idx = np.arange(len(self))[key]
        else:
            idx = np.asarray(key)
            if idx.dtype == bool:
                idx = np.nonzero(idx)[0]
        return CellData(self.X[idx], self.obs.iloc[idx], self.var_names)

    def copy(self):
        return CellData(self.X.copy(), self.obs.copy(), self.var_names.copy())

    def __repr__(self):
        return f"CellData object with n_obs x n_vars = {self.n_obs} x {self.n_vars}\n    obs: {list(self.obs.columns)}"
