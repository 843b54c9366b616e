"""Product surfaces pinned by outputs of the REAL reference (tests/golden/make_goldens_round2.py):

  * a checkpoint directory written by the reference's BaseMixin.save (models/base/_base.py:42-90) loads into
    scarches_amd.models.TRVAE and reproduces the reference's own get_latent(mean=True) / get_y;
  * scPoli.classify (prototype=True and on cells, distance / probability forms, scaled uncertainties),
    get_recon_loss (own and substituted condition encoders) and add_new_cell_type against the real scPoli wrapper
    (models/scpoli/scpoli.py:213-315, scpoli_model.py:411-519,919-947), reparameterisation noise off on both sides;
  * scPoli.shot_surgery (scpoli_model.py:842-917): substitution losses, mapping and copied parameters;
  * EXPIMAP.latent_enrich Bayes factors (models/expimap/expimap_model.py:445-591, exact=True) under the same numpy seed.
"""
import contextlib
import io
import os

import numpy as np
import pandas as pd
import pytest
import torch

from tests._util import GOLDEN

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_reference_written_checkpoint_loads_and_maps_identically():
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import TRVAE
    z = np.load(os.path.join(GOLDEN, "ref_ckpt_trvae.npz"))
    d = os.path.join(GOLDEN, "ref_ckpt_trvae")
    assert sorted(os.listdir(d)) == ["attr.pkl", "model_params.pt", "var_names.csv"]
    ad = CellData(z["X"], pd.DataFrame({"study": z["study"].astype(str)}), var_names=z["genes"].astype(str))
    m = quiet(TRVAE.load, d, ad)
    assert m.is_trained_ and list(m.conditions_) == [str(s) for s in z["conditions"]]
    assert not m.model.training
    zm = m.get_latent(ad.X, ad.obs["study"], mean=True)
    assert zm.dtype == np.float32 and zm.shape == z["latent_mean"].shape
    assert np.max(np.abs(zm - z["latent_mean"])) <= 1e-5 * np.max(np.abs(z["latent_mean"]))
    # (get_y decodes a SAMPLED z in the reference, _base.py:434-455: its stored output is one random draw - only the
    #  shape and the ReLU range are comparable)
    y = m.get_y(ad.X, ad.obs["study"])
    assert y.shape == z["y"].shape and (y >= 0).all()
    # genes in another order (and one missing): load re-orders / zero-fills to the saved var names like the reference
    perm = np.random.default_rng(0).permutation(len(z["genes"]))
    ad2 = CellData(z["X"][:, perm], pd.DataFrame({"study": z["study"].astype(str)}),
                   var_names=z["genes"].astype(str)[perm])
    m2 = quiet(TRVAE.load, d, ad2)
    zm2 = m2.get_latent(mean=True)
    assert np.max(np.abs(zm2 - z["latent_mean"])) <= 1e-5 * np.max(np.abs(z["latent_mean"]))
    # ... and is the starting point of surgery, straight from the reference's files
    q = quiet(TRVAE.load_query_data, ad, d)
    assert [n for n, p in q.model.named_parameters() if p.requires_grad] == [
        "theta", "encoder.FC.L0.cond_L.weight", "decoder.FirstL.L0.cond_L.weight"]


@pytest.fixture(scope="module")
def scpoli_model():
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import scPoli
    z = np.load(os.path.join(GOLDEN, "scpoli_api.npz"))
    obs = pd.DataFrame({"study": z["study"].astype(str), "cell_type": z["cell_type"].astype(str)})
    ad = CellData(z["X"], obs)
    torch.manual_seed(7)
    m = quiet(scPoli, ad, condition_keys="study", cell_type_keys="cell_type", hidden_layer_sizes=[20], latent_dim=5,
              embedding_dims=4, recon_loss="nb", dr_rate=0.0)
    assert list(m.conditions_["study"]) == [str(s) for s in z["conditions"]]
    assert list(m.cell_types_.keys()) == [str(s) for s in z["cell_types"]]
    sd = {k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w/")}
    # same torch seed -> the reference's initial weights, bit for bit
    for k, v in m.model.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    m.model.load_state_dict(sd)
    m.model.eval()
    m.model.engine().noise_off = True
    protos = {"mean": torch.tensor(z["protos"]).cuda(), "cov": torch.tensor(z["protos_cov"]).cuda()}
    pu = {"mean": torch.tensor(z["protos_u"]).cuda()}
    m.prototypes_labeled_, m.prototypes_unlabeled_ = protos, pu
    m.model.prototypes_labeled, m.model.prototypes_unlabeled = protos, pu
    return m, ad, z


def test_scpoli_classify_matches_reference(scpoli_model):
    m, ad, z = scpoli_model
    for tag, kw in (("dist", dict(get_prob=False)), ("prob", dict(get_prob=True))):
        r = m.classify(z["cls_lat"], prototype=True, **kw)["cell_type"]
        assert list(r["preds"]) == [str(s) for s in z[f"cls_lat_{tag}_preds"]], tag
        assert np.allclose(r["uncert"], z[f"cls_lat_{tag}_uncert"], rtol=1e-5, atol=1e-6), tag
        assert np.allclose(r["weighted_distances"], z[f"cls_lat_{tag}_wd"], rtol=1e-5, atol=1e-6), tag
    r = m.classify(ad)["cell_type"]
    wd_ref = z["cls_cells_wd"]
    assert np.allclose(r["weighted_distances"], wd_ref, rtol=2e-5, atol=2e-6)
    assert np.allclose(r["uncert"], z["cls_cells_uncert"], rtol=2e-5, atol=2e-6)
    # predictions agree wherever the two nearest prototypes are not tied to rounding
    srt = np.sort(wd_ref, 1)
    clear = (srt[:, 1] - srt[:, 0]) > 1e-4
    assert clear.mean() > 0.9
    assert (r["preds"][clear] == z["cls_cells_preds"].astype(str)[clear]).all()
    r2 = m.classify(ad, scale_uncertainties=True)["cell_type"]
    assert np.allclose(r2["uncert"], z["cls_cells_uncert_scaled"], rtol=1e-4, atol=1e-5)


def test_scpoli_get_recon_loss_matches_reference(scpoli_model):
    from copy import deepcopy
    m, ad, z = scpoli_model
    sub = ad[z["recon_rows"]]
    own = m.get_recon_loss(sub, batch_size=16)
    assert abs(own - float(z["recon_loss_own"])) <= 1e-5 * abs(float(z["recon_loss_own"]))
    ce, cce = deepcopy(m.model.condition_encoders), deepcopy(m.model.conditions_combined_encoder)
    ce["study"]["sC"] = ce["study"]["sA"]
    cce["sC"] = cce["sA"]
    sub_loss = m.get_recon_loss(sub, 16, ce, cce)
    assert abs(sub_loss - float(z["recon_loss_subst"])) <= 1e-5 * abs(float(z["recon_loss_subst"]))
    assert abs(sub_loss - own) > 1.0          # the substitution matters


def test_scpoli_add_new_cell_type_matches_reference(scpoli_model):
    m, ad, z = scpoli_model
    m.add_new_cell_type("ct_new", "cell_type", [0, 2])
    assert list(m.cell_types_.keys()) == [str(s) for s in z["new_cell_types"]]
    mean = m.model.prototypes_labeled["mean"].cpu().numpy()
    cov = m.model.prototypes_labeled["cov"].cpu().numpy()
    assert mean.shape == z["new_protos"].shape and cov.shape == z["new_protos_cov"].shape
    assert np.allclose(mean, z["new_protos"], rtol=1e-6, atol=1e-7)
    assert np.allclose(cov, z["new_protos_cov"], rtol=1e-4, atol=1e-5)


def test_scpoli_shot_surgery_matches_reference():
    """scPoli.shot_surgery (scpoli_model.py:842-917, train_epochs = 0) against the real wrapper: the reconstruction loss
    of every (new study, reference study) substitution as printed, the chosen mapping, and the embedding rows / theta
    columns copied into the returned model."""
    import re
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.models import scPoli
    z = np.load(os.path.join(GOLDEN, "scpoli_shot.npz"))
    ad = CellData(z["X"], pd.DataFrame({"study": z["study"].astype(str), "cell_type": z["cell_type"].astype(str)}))
    torch.manual_seed(11)
    ref = quiet(scPoli, ad, condition_keys="study", cell_type_keys="cell_type", hidden_layer_sizes=[20], latent_dim=5,
                embedding_dims=4, recon_loss="nb", dr_rate=0.0)
    ref.model.load_state_dict({k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w/")})
    adq = CellData(z["Xq"], pd.DataFrame({"study": z["study_q"].astype(str),
                                          "cell_type": z["cell_type_q"].astype(str)}))
    buf = io.StringIO()
    CVAEEngine.noise_off = True            # the wrapper builds the query model itself: switch the noise off class-wide
    try:
        torch.manual_seed(12)
        with contextlib.redirect_stdout(buf):
            model, mapping = scPoli.shot_surgery(adq, ref, labeled_indices=list(range(len(z["Xq"]))), train_epochs=0,
                                                 batch_size=16)
    finally:
        CVAEEngine.noise_off = False
    assert list(model.conditions_["study"]) == [str(s) for s in z["conditions"]]
    want = dict(zip((str(s) for s in z["loss_keys"]), z["loss_vals"]))
    got, cur = {}, None
    for line in buf.getvalue().splitlines():
        m1 = re.match(r"Processing (\S+)\.", line.strip())
        m2 = re.match(r"Using (\S+), recon loss ([-0-9.e+]+)\.", line.strip())
        if m1:
            cur = m1.group(1)
        elif m2:
            got[f"{cur}|{m2.group(1)}"] = float(m2.group(2))
    assert set(got) == set(want) and len(want) == 6
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-5 * abs(v), (k, got[k], v)
    assert mapping == dict(zip((str(s) for s in z["map_new"]), (str(s) for s in z["map_old"])))
    emb = model.model.embeddings[0].weight.detach().cpu().numpy()
    theta = model.model.theta.detach().cpu().numpy()
    # (the forward passes of get_recon_loss renormalise the touched embedding rows in place - nn.Embedding(max_norm),
    #  scpoli.py:99-101 - which the HIP kernel does to the last bit or one ulp off: compare to 1e-6, copies exactly)
    assert np.allclose(emb, z["emb_after"], rtol=1e-6, atol=1e-7) and np.allclose(theta, z["theta_after"], rtol=1e-6)
    enc = model.model.condition_encoders["study"]
    comb = model.model.conditions_combined_encoder
    for new, old in mapping.items():
        assert np.array_equal(emb[enc[new]], emb[enc[old]]) and np.array_equal(theta[:, comb[new]], theta[:, comb[old]])
    # the query model maps its cells with the borrowed embeddings right away
    lat = model.get_latent(adq, mean=True)
    assert lat.shape == (len(z["Xq"]), 5) and np.isfinite(lat).all()


def test_expimap_latent_enrich_matches_reference():
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import EXPIMAP
    z = np.load(os.path.join(GOLDEN, "expimap_enrich.npz"))
    ad = CellData(z["X"], pd.DataFrame({"study": z["study"].astype(str), "group": z["group"].astype(str)}))
    torch.manual_seed(9)
    m = quiet(EXPIMAP, ad, condition_key="study", conditions=["r0", "r1"], hidden_layer_sizes=[24], mask=z["mask"],
              recon_loss="nb")
    sd = {k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w/")}
    m.model.load_state_dict(sd)
    m.model.eval()
    np.random.seed(123)
    m.latent_enrich("group", comparison="rest", n_sample=200, exact=True, key_added="bf_rest")
    np.random.seed(124)
    m.latent_enrich("group", comparison="ctrl", n_sample=150, exact=True, key_added="bf_ctrl", select_terms=[0, 2, 5])
    n = 0
    for key in z.files:
        if not (key.startswith("rest/") or key.startswith("ctrl/")):
            continue
        which, cat, stat = key.split("/")
        got = np.asarray(ad.uns["bf_" + which][cat][stat], dtype=np.float64)
        ref = z[key]
        assert got.shape == ref.shape, key
        # probabilities are means of erfc terms of fp32 latent statistics; the Bayes factor is a log-odds of them
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-5), (key, got, ref)
        n += 1
    assert n == 3 * 3 + 2 * 3
