"""CPU suite: the scPoli oracle (oracle/scpoli_oracle.py) against fixtures produced by the REAL reference
(tests/golden/make_goldens_scpoli.py).  This is what pins that oracle."""
import numpy as np
import pytest
import torch

from oracle import cvae_oracle as O
from oracle import scpoli_oracle as S
from tests._util import GOLDEN, dense_from_case, golden_ref64, grads_close, load_case, rel_err

torch.set_num_threads(1)
TOL = 1e-5
SCPOLI_CASES = ["scpoli_nb_two_cov", "scpoli_nb_deep", "scpoli_mse_dec_only", "scpoli_zinb", "scpoli_poisson"]


def run(z, cfg, W, masks, P=None):
    x = dense_from_case(z, cfg["G"])
    return S.scpoli_loss_and_grads(P if P is not None else W, cfg, x, z["batch"], z["combined"],
                                   torch.tensor(z["sizefactor"]), torch.tensor(z["eps"]), float(z["alpha"]),
                                   float(z["eta"]), float(z["unlabeled_weight"]), z["celltypes"], z["labeled"],
                                   z["protos"], z["protos_u"], masks or None)


@pytest.mark.parametrize("name", SCPOLI_CASES)
def test_scpoli_losses_and_grads(name):
    z, cfg, W, Gr, masks = load_case(name)
    res, grads, aux = run(z, cfg, W, masks)
    for k, ref in (("recon", "recon"), ("kl", "kl"), ("labeled", "labeled_loss"), ("unlabeled", "unlabeled"),
                   ("loss", "loss")):
        assert abs(res[k] - float(z[ref])) <= TOL * abs(float(z[ref])), k
    assert rel_err(aux["z"].numpy(), z["latent"]) < TOL
    bad = grads_close({k: v.numpy() for k, v in grads.items()}, Gr, rtol=1e-5, ref64=golden_ref64(name)[1])
    assert not bad, bad
    for k in W:                                # in-place max_norm renormalisation of the touched embedding rows
        if "embedding" in k:
            assert rel_err(aux["P"][k].detach().numpy(), z["w_after/" + k]) < 1e-6, k


@pytest.mark.parametrize("name", ["scpoli_nb_two_cov", "scpoli_nb_deep", "scpoli_mse_dec_only", "scpoli_poisson"])
def test_scpoli_adam_two_groups(name):
    """Adam with the embeddings in their own weight_decay=0 group (trainers/scpoli/trainer.py:288-305)."""
    z, cfg, W, _, masks = load_case(name)
    steps, wd = int(z["adam_steps"]), float(z["adam_wd"])
    P = {k: torch.tensor(v.copy()) for k, v in W.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    # the golden's first forward (before its Adam loop) already renormalised the embeddings
    _, _, aux = run(z, cfg, W, masks)
    for k in P:
        if "embedding" in k:
            P[k] = aux["P"][k].detach().clone()
    for t in range(1, steps + 1):
        _, grads, aux = run(z, cfg, W, masks, P={k: v.numpy() for k, v in P.items()})
        for k in P:
            if "embedding" in k:
                P[k] = aux["P"][k].detach().clone()
            O.adam_step(P[k], grads[k], M[k], V[k], t, lr=1e-3, eps=0.01, wd=0.0 if "embedding" in k else wd)
    for k in P:
        ref = z["adam/" + k]
        assert np.max(np.abs(P[k].numpy() - ref)) <= 3e-6 * max(1.0, np.max(np.abs(ref))), k


def test_scpoli_eval_latent():
    z, cfg, W, _, _ = load_case("scpoli_nb_two_cov")
    P = {k: torch.tensor(v.copy()) for k, v in W.items()}
    x = dense_from_case(z, cfg["G"])
    _, _, _, aux = S.scpoli_forward(P, cfg, x, torch.tensor(z["batch"]), torch.tensor(z["combined"]),
                                    torch.tensor(z["sizefactor"]), torch.tensor(z["eps"]), train=False)
    assert rel_err(aux["mu"].numpy(), z["z_mean"]) < TOL


def test_prototype_fixtures():
    z = np.load(GOLDEN + "/scpoli_proto.npz")
    lat, labels = torch.tensor(z["up_latent"]), torch.tensor(z["up_labels"])
    pm, pc = torch.tensor(z["up_prev_mean"]), torch.tensor(z["up_prev_cov"])
    m, c = S.update_labeled_prototypes(lat, labels, 4, pm, pc, None)
    assert rel_err(m.numpy(), z["up_mean"]) < 1e-6 and rel_err(c.numpy(), z["up_cov"]) < 1e-6
    m, c = S.update_labeled_prototypes(lat, labels, 4, pm, pc, [0, 3])
    assert rel_err(m.numpy(), z["up_mask_mean"]) < 1e-6 and rel_err(c.numpy(), z["up_mask_cov"]) < 1e-6
    P = torch.tensor(z["ul_P"], requires_grad=True)
    lg = lat.clone().requires_grad_(True)
    v, counts = S.prototype_unlabeled_loss(lg, P)
    v.backward()
    assert abs(v.item() - float(z["ul_val"])) <= 1e-6 * float(z["ul_val"])
    assert np.array_equal(counts.numpy(), z["ul_counts"])
    assert rel_err(lg.grad.numpy(), z["ul_dlat"]) < 1e-5 and rel_err(P.grad.numpy(), z["ul_dP"]) < 1e-5
    lg = lat.clone().requires_grad_(True)
    v = S.prototype_labeled_loss(lg, pm, labels)
    v.backward()
    assert abs(v.item() - float(z["ll_val"])) <= 1e-6 * float(z["ll_val"])
    assert rel_err(lg.grad.numpy(), z["ll_dlat"]) < 1e-5


def test_multi_condition_dataset_matches_reference():
    """Host logic: MultiConditionAnnotatedDataset encodings / items against the reference's (dataset/scpoli)."""
    import pandas as pd
    import scipy.sparse as sp
    import scarches_amd as sca
    z = np.load(GOLDEN + "/scpoli_dataset.npz")
    n, g = z["shape"]
    xs = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(n, g))
    obs = pd.DataFrame({"study": z["study"].astype(str), "tech": z["tech"].astype(str), "ct": z["ct"].astype(str)})
    obs["conditions_combined"] = obs[["study", "tech"]].apply(lambda r: "_".join(r), axis=1)
    ad = sca.CellData(xs, obs)
    enc = {"study": {"s0": 0, "s1": 1, "s2": 2}, "tech": {"a": 0, "b": 1}}
    comb = {c: i for i, c in enumerate(z["combined_names"].astype(str).tolist())}
    lab = np.zeros((n, 1))
    lab[::2] = 1
    ds = sca.dataset.MultiConditionAnnotatedDataset(ad, ["study", "tech"], enc, comb, ["ct"],
                                                    {"T": 0, "B": 1, "NK": 2, "unk": -1}, lab)
    assert np.array_equal(ds.conditions.numpy(), z["conditions"])
    assert np.array_equal(ds.conditions_combined.numpy(), z["conditions_combined"])
    assert np.array_equal(ds.cell_types.numpy(), z["cell_types"])
    assert np.allclose(ds.size_factors.numpy(), z["size_factors"])
    assert np.array_equal(ds.labeled_vector.numpy(), z["labeled"])
    item = ds[np.array([0, 5, 7])]
    assert np.array_equal(item["x"].numpy(), z["item_x"])
    assert np.array_equal(item["batch"].numpy(), z["item_batch"])
    assert np.array_equal(item["combined_batch"].numpy(), z["item_combined"])
    assert np.array_equal(item["celltypes"].numpy(), z["item_celltypes"])
