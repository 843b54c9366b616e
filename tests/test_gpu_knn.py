"""GPU parity of the weighted-kNN label transfer (sca_knn_topk / sca_knn_vote, scarches_amd.utils.knn) against the REAL
reference's outputs (tests/golden/knn.npz), the oracle, and - at a size the oracle does not reach - torch.cdist+topk."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn.npz")
CASES = ["basic", "dup", "const", "twocol", "unknown"]


class Emb:
    """Duck-typed AnnData: X, obsm, obs_names (what scarches/utils/knn.py touches)."""

    def __init__(self, X):
        self.X = X
        self.obsm = {"X_emb": X}
        self.obs_names = pd.Index([f"c{i}" for i in range(X.shape[0])])


def columns(z, name):
    return [k[len(name) + 5:] for k in z.files if k.startswith(name + "_obs_")]


@pytest.mark.parametrize("name", CASES)
def test_topk_matches_reference(name):
    from scarches_amd.utils.knn import HipKNeighbors
    z = np.load(GOLDEN)
    k = int(z[name + "_k"])
    ref, qry = z[name + "_ref"], z[name + "_qry"]
    dist, idx = HipKNeighbors(ref, k).kneighbors(qry)
    gd = z[name + "_dist"]
    # fp32 sum of squared differences vs sklearn's float64 |x|^2+|y|^2-2xy: rel 1e-6, abs 1e-6 at coincident points
    assert np.all(np.abs(dist - gd) <= 2e-6 * np.maximum(1.0, gd)), np.abs(dist - gd).max()
    assert np.all(np.diff(dist, axis=1) >= 0)
    # every returned index really lies at the reported distance (ties may be listed in another order)
    true = np.sqrt(((qry[:, None, :].astype(np.float64) - ref[idx].astype(np.float64)) ** 2).sum(-1))
    assert np.all(np.abs(true - dist) <= 2e-6 * np.maximum(1.0, true))
    for row in idx:
        assert len(set(row.tolist())) == k


@pytest.mark.parametrize("name", CASES)
def test_vote_matches_reference(name):
    from scarches_amd import _lib
    lib = _lib.load()
    z = np.load(GOLDEN)
    k = int(z[name + "_k"])
    dist = torch.tensor(z[name + "_dist"], dtype=torch.float32, device="cuda").contiguous()
    idx = torch.tensor(z[name + "_idx"], dtype=torch.int32, device="cuda").contiguous()
    nq = dist.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    for col in columns(z, name):
        uniq, codes = np.unique(z[f"{name}_obs_{col}"], return_inverse=True)
        lab = torch.tensor(codes.astype(np.int32), device="cuda")
        pred = torch.empty(nq, dtype=torch.int32, device="cuda")
        prob = torch.empty(nq, dtype=torch.float64, device="cuda")
        _lib.check(lib.sca_knn_vote(dist.data_ptr(), idx.data_ptr(), nq, k, lab.data_ptr(), pred.data_ptr(),
                                    prob.data_ptr(), st), "vote")
        torch.cuda.synchronize()
        unc = np.maximum(1 - prob.cpu().numpy(), 0)
        got = uniq[pred.cpu().numpy()].astype(str)
        if name == "unknown":
            got = np.where(prob.cpu().numpy() >= 0.8, got, "Unknown")
        gold = z[f"{name}_pred_{col}"]
        # distances enter as fp32: a label may only differ where two candidates are tied to ~1e-6
        diff = got != gold
        assert diff.mean() <= 0.01, diff.mean()
        assert np.abs(unc - z[f"{name}_unc_{col}"])[~diff].max() <= 1e-5


@pytest.mark.parametrize("name", ["basic", "twocol", "unknown", "const"])
def test_api_matches_reference(name):
    from scarches_amd.utils.knn import weighted_knn_trainer, weighted_knn_transfer
    z = np.load(GOLDEN)
    k = int(z[name + "_k"])
    cols = columns(z, name)
    obs = pd.DataFrame({c: z[f"{name}_obs_{c}"] for c in cols})
    emb = "X_emb" if name == "twocol" else "X"
    model = weighted_knn_trainer(Emb(z[name + "_ref"]), emb, n_neighbors=k)
    kw = dict(threshold=0.8, pred_unknown=True) if name == "unknown" else {}
    pred, unc = weighted_knn_transfer(Emb(z[name + "_qry"]), emb, obs, "cell_type", model, **kw)
    assert list(pred.columns) == [c for c in obs.columns if c.startswith("cell_type")]
    for c in cols:
        same = pred[c].to_numpy().astype(str) == z[f"{name}_pred_{c}"]
        assert same.mean() >= 0.99
        assert np.abs(unc[c].to_numpy().astype(float) - z[f"{name}_unc_{c}"])[same].max() <= 1e-5
    with pytest.raises(ValueError, match="should be set to either 'X'"):
        weighted_knn_trainer(Emb(z[name + "_ref"]), "nope", n_neighbors=k)
    with pytest.raises(ValueError, match="knn_model should be of type"):
        weighted_knn_transfer(Emb(z[name + "_qry"]), emb, obs, "cell_type", object())


def test_topk_large_against_torch():
    """200k reference cells x 4096 queries, Z = 10, k = 50: distances and neighbour sets vs torch.cdist + topk."""
    from scarches_amd.utils.knn import HipKNeighbors
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    ref = torch.randn(200_000, 10, device="cuda", generator=gen)
    qry = torch.randn(4096, 10, device="cuda", generator=gen)
    model = HipKNeighbors(ref.cpu().numpy(), 50)
    dist, idx = model.kneighbors_device(qry)
    d = torch.cdist(qry.double(), ref.double())
    td, ti = torch.topk(d, 50, dim=1, largest=False)
    torch.cuda.synchronize()
    assert float((dist.double() - td).abs().max()) <= 2e-6 * float(td.max())
    # continuous random data: no ties, the index lists must agree except for fp32-vs-fp64 swaps of near-equal pairs
    agree = (idx.long() == ti).float().mean().item()
    assert agree >= 0.999, agree
