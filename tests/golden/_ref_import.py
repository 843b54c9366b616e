"""Import harness for the REAL reference (theislab/scarches @ /root/reference).

Runs ONLY in the build container (the reference never travels to the GPU box).
Used by tests/golden/make_goldens.py to emit fixtures; nothing in the product
path or in the tests imports this module.

The reference package cannot be imported normally here (anndata / scanpy /
scvi-tools are not installed), so namespace modules that point at the reference
directories are registered and the few missing third-party names are stubbed.
Recipe: SURVEY.md Appendix C.
"""
import sys
import types

import numpy as np
import pandas as pd

REF = '/root/reference/scarches'


class FakeAnnData:
    """Duck-typed AnnData stand-in: X (ndarray | scipy csr), obs (DataFrame), var_names."""

    def __init__(self, X, obs=None, var_names=None, dtype=None):
        self.X = X
        n = X.shape[0]
        self.obs = (obs.reset_index(drop=True) if obs is not None
                    else pd.DataFrame(index=np.arange(n)))
        self.var_names = pd.Index(var_names if var_names is not None
                                  else [f"g{i}" for i in range(X.shape[1])])
        self.varm = {}
        self.uns = {}

    shape = property(lambda s: s.X.shape)
    n_vars = property(lambda s: s.X.shape[1])
    n_obs = property(lambda s: s.X.shape[0])

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, i):
        if isinstance(i, tuple):
            rows, cols = i
            cols = self.var_names.get_indexer(pd.Index(cols))
            ad = self if isinstance(rows, slice) and rows == slice(None) else self[rows]
            return FakeAnnData(ad.X[:, cols], ad.obs, self.var_names[cols])
        i = np.asarray(i)
        obs = self.obs[i] if i.dtype == bool else self.obs.iloc[i]
        return FakeAnnData(self.X[i], obs, self.var_names)

    def copy(self):
        return FakeAnnData(self.X.copy(), self.obs.copy(), self.var_names.copy())


def install():
    def ns(name, sub):
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m

    for name, sub in [('scarches', ''), ('scarches.models', '/models'),
                      ('scarches.models.trvae', '/models/trvae'),
                      ('scarches.models.base', '/models/base'),
                      ('scarches.models.expimap', '/models/expimap'),
                      ('scarches.models.scpoli', '/models/scpoli'),
                      ('scarches.trainers', '/trainers'),
                      ('scarches.trainers.trvae', '/trainers/trvae'),
                      ('scarches.trainers.expimap', '/trainers/expimap'),
                      ('scarches.trainers.scpoli', '/trainers/scpoli'),
                      ('scarches.dataset', '/dataset'),
                      ('scarches.dataset.trvae', '/dataset/trvae'),
                      ('scarches.dataset.scpoli', '/dataset/scpoli'),
                      ('scarches.utils', '/utils')]:
        ns(name, sub)
    for modname, attrs in [('scvi', {}), ('scvi.distributions', {'NegativeBinomial': object}),
                           ('anndata', {'AnnData': FakeAnnData, 'read': None}),
                           ('scanpy', {'AnnData': FakeAnnData})]:
        m = types.ModuleType(modname)
        m.__dict__.update(attrs)
        sys.modules[modname] = m
    import scarches.dataset.trvae.anndata as _t
    import scarches.dataset.scpoli.anndata as _s
    sys.modules['scarches.dataset'].trVAEDataset = _t.AnnotatedDataset
    sys.modules['scarches.dataset'].MultiConditionAnnotatedDataset = _s.MultiConditionAnnotatedDataset
    import scarches.trainers.expimap.regularized as _r
    import scarches.trainers.scpoli.trainer as _st
    sys.modules['scarches.trainers'].expiMapTrainer = _r.expiMapTrainer
    sys.modules['scarches.trainers.scpoli'].scPoliTrainer = _st.scPoliTrainer
