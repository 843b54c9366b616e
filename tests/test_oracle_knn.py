"""The kNN label-transfer oracle against the REAL reference's outputs (tests/golden/knn.npz)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn.npz")
CASES = ["basic", "dup", "const", "twocol", "unknown"]


def columns(z, name):
    return [k[len(name) + 5:] for k in z.files if k.startswith(name + "_obs_")]


@pytest.mark.parametrize("name", CASES)
def test_kneighbors_and_vote_match_reference(name):
    from oracle import knn_oracle as O
    z = np.load(GOLDEN)
    k = int(z[name + "_k"])
    dist, idx = O.kneighbors(z[name + "_ref"], z[name + "_qry"], k)
    # sklearn evaluates |x|^2 + |y|^2 - 2xy: absolute 1e-6 for coincident points, rel 1e-6 otherwise
    assert np.all(np.abs(dist - z[name + "_dist"]) <= 1e-6 * np.maximum(1.0, z[name + "_dist"]))
    kw = dict(threshold=0.8, pred_unknown=True) if name == "unknown" else {}
    for col in columns(z, name):
        # vote on the reference's own neighbour lists (ties at equal distance may be ordered differently)
        pred, unc = O.weighted_knn_vote(z[name + "_dist"], z[name + "_idx"], z[f"{name}_obs_{col}"], **kw)
        assert list(pred.astype(str)) == list(z[f"{name}_pred_{col}"])
        assert np.allclose(unc, z[f"{name}_unc_{col}"], rtol=0, atol=1e-12)
        # and on the oracle's own lists: same labels wherever the winning margin is not a rounding tie
        pred2, unc2 = O.weighted_knn_vote(dist, idx, z[f"{name}_obs_{col}"], **kw)
        same = pred2.astype(str) == z[f"{name}_pred_{col}"]
        assert same.mean() >= 0.99 and np.abs(unc2 - z[f"{name}_unc_{col}"])[same].max() <= 1e-5
