"""Golden vectors for the weighted-kNN label transfer by RUNNING THE REAL REFERENCE
(scarches/utils/knn.py:10-139 weighted_knn_trainer / weighted_knn_transfer, sklearn KNeighborsTransformer underneath).
Build container only; writes tests/golden/knn.npz (inputs + the reference's outputs, no source text)."""
import contextlib
import io
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
from _ref_import import FakeAnnData  # noqa: E402
from scarches.utils.knn import weighted_knn_trainer, weighted_knn_transfer  # noqa: E402


class Emb(FakeAnnData):
    def __init__(self, X, obs=None):
        super().__init__(X, obs)
        self.obsm = {"X_emb": X}
        self.obs_names = pd.Index([f"c{i}" for i in range(X.shape[0])])


def case(rng, nr, nq, d, n_lab, dup=False, const=False, two_cols=False):
    centers = rng.normal(0, 2.0, size=(n_lab, d))
    yr = rng.integers(0, n_lab, size=nr)
    ref = (centers[yr] + rng.normal(0, 1.0, size=(nr, d))).astype(np.float32)
    yq = rng.integers(0, n_lab, size=nq)
    qry = (centers[yq] + rng.normal(0, 1.0, size=(nq, d))).astype(np.float32)
    if dup:                      # exact duplicates: zero distances and distance ties
        ref[nr // 2:] = ref[:nr - nr // 2]
        qry[: nq // 4] = ref[: nq // 4]
    if const:                    # every neighbour at the same distance: std = 0 -> uniform weights
        ref[:] = ref[0]
    names = np.array([f"type_{i:02d}" for i in range(n_lab)])
    obs = pd.DataFrame({"cell_type": names[yr]})
    if two_cols:
        obs["cell_type_coarse"] = np.array(["A", "B", "C"])[yr % 3]
    return ref, qry, obs


def run(ref, qry, obs, k, emb="X", **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        model = weighted_knn_trainer(Emb(ref), emb, n_neighbors=k)
        dist, idx = model.kneighbors(X=qry)
        pred, unc = weighted_knn_transfer(Emb(qry), emb, obs, "cell_type", model, **kw)
    return dist, idx, pred, unc


def main():
    rng = np.random.default_rng(7)
    out = {}
    specs = {
        "basic": dict(nr=3000, nq=400, d=10, k=50, n_lab=7),
        "dup": dict(nr=800, nq=200, d=10, k=50, n_lab=5, dup=True),
        "const": dict(nr=120, nq=16, d=4, k=50, n_lab=3, const=True),
        "twocol": dict(nr=1500, nq=150, d=16, k=15, n_lab=9, two_cols=True),
        "unknown": dict(nr=2000, nq=300, d=10, k=50, n_lab=12),
    }
    for name, sp in specs.items():
        k = sp.pop("k")
        ref, qry, obs = case(rng, **sp)
        kw = dict(threshold=0.8, pred_unknown=True) if name == "unknown" else {}
        emb = "X_emb" if name == "twocol" else "X"
        with np.errstate(all="ignore"):
            dist, idx, pred, unc = run(ref, qry, obs, k, emb, **kw)
        out[f"{name}_ref"], out[f"{name}_qry"], out[f"{name}_k"] = ref, qry, np.int64(k)
        out[f"{name}_dist"], out[f"{name}_idx"] = dist, idx
        for col in obs.columns:
            out[f"{name}_obs_{col}"] = obs[col].to_numpy().astype(str)
            out[f"{name}_pred_{col}"] = pred[col].to_numpy().astype(str)
            out[f"{name}_unc_{col}"] = unc[col].to_numpy().astype(np.float64)
        print(name, dist.dtype, dist.shape, pred.iloc[:3].to_dict("records"), unc.iloc[:3].to_dict("records"))
    np.savez_compressed(os.path.join(HERE, "knn.npz"), **out)


if __name__ == "__main__":
    main()
