import logging

import numpy as np
import scipy.sparse as sp

from ...anndata_lite import CellData

logger = logging.getLogger(__name__)


def _validate_var_names(adata, source_var_names):
    """Bring the query genes into the reference order (scarches/models/base/_utils.py:13-63): genes of the reference
    that the query lacks are appended as all-zero columns, genes unknown to the reference are dropped, and the
    columns are re-ordered to `source_var_names`.  Returns `adata` itself when nothing has to change."""
    user = adata.var_names.astype(str)
    source = [str(s) for s in np.asarray(source_var_names).tolist()]
    shared = len(set(user).intersection(source))
    try:
        pct = round(shared / len(user) * 100, 4)
        if pct != 100:
            logger.warning(f"WARNING: Query shares {pct}% of its genes with the reference."
                           "This may lead to inaccuracy in the results.")
    except Exception:
        logger.warning("WARNING: Something is wrong with the reference genes.")

    user_set = set(user)
    missing = [g for g in source if g not in user_set]
    out = adata
    if missing:
        print("Query data is missing expression data of ", len(missing),
              " genes which were contained in the reference dataset.")
        print("The missing information will be filled with zeroes.")
        if sp.issparse(adata.X):
            X = sp.hstack((adata.X.tocsr(), sp.csr_matrix((len(adata), len(missing)), dtype=adata.X.dtype))).tocsr()
        else:
            X = np.concatenate((np.asarray(adata.X), np.zeros((len(adata), len(missing)))), axis=1)
        out = CellData(X, adata.obs.copy(), list(user) + missing, dtype="float32")
    extra = len(user) - (len(source) - len(missing))
    if extra > 0:
        print("Query data contains expression data of ", extra,
              " genes that were not contained in the reference dataset. This information "
              "will be removed from the query data object for further processing.")
    if list(out.var_names.astype(str)) != source:
        out = out[:, source].copy()
    return out
