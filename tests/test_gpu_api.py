"""End-to-end GPU tests through the reference's public API: TRVAE(...).train(), get_latent(), save(),
load_query_data() surgery with frozen weights, and the acceptance check of the north star: latent kNN label-transfer
accuracy of the HIP-trained model against a model trained by the CPU oracle on the same data."""
import contextlib
import io

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def synth(n, G, n_cond, n_types, seed, cond_offset=0):
    rng = np.random.default_rng(seed)
    prof = rng.gamma(0.25, 1.0, size=(n_types, G))
    prof /= prof.sum(1, keepdims=True)
    shift = np.exp(rng.normal(0, 0.35, size=(n_cond, G)))
    t = rng.integers(0, n_types, size=n)
    c = rng.integers(0, n_cond, size=n)
    s = 400 * np.exp(rng.normal(0, 0.3, size=n))
    mean = s[:, None] * prof[t] * shift[c]
    mean *= s[:, None] / mean.sum(1, keepdims=True)
    x = rng.poisson(rng.gamma(2.0, mean / 2.0)).astype(np.float32)
    obs = pd.DataFrame({"study": [f"s{cond_offset + i}" for i in c], "cell_type": [f"t{i}" for i in t]})
    return sp.csr_matrix(x), obs


def knn_acc(train_z, train_y, test_z, test_y, k=15):
    from sklearn.neighbors import KNeighborsClassifier
    clf = KNeighborsClassifier(n_neighbors=k, weights="distance").fit(train_z, train_y)
    return float((clf.predict(test_z) == test_y).mean())


@pytest.fixture(scope="module")
def trained():
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import TRVAE
    # the query shares the cell-type programmes of the reference (same seed) but comes from two new studies
    X, obs = synth(3000, 256, 5, 5, seed=3)
    is_q = obs["study"].isin(["s3", "s4"]).to_numpy()
    ad = CellData(X[~is_q], obs[~is_q])
    adq = CellData(X[is_q], obs[is_q])
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(0)
        m = TRVAE(ad, condition_key="study", hidden_layer_sizes=[64, 32], latent_dim=8, dr_rate=0.05)
        init = {k: v.clone() for k, v in m.model.state_dict().items()}
        m.train(n_epochs=30, lr=1e-3, eps=0.01, batch_size=128, alpha_epoch_anneal=20, alpha_kl=0.5,
                early_stopping_kwargs=dict(patience=50), monitor=False)
    return m, ad, adq, init


def test_train_logs_and_latent(trained):
    m, ad, adq, init = trained
    logs = m.trainer.logs
    for k in ("epoch_loss", "epoch_unweighted_loss", "epoch_recon_loss", "epoch_kl_loss", "epoch_mmd_loss",
              "val_loss", "val_unweighted_loss", "val_recon_loss", "val_kl_loss", "val_mmd_loss"):
        assert len(logs[k]) == 30 and np.isfinite(logs[k]).all(), k
    assert logs["epoch_recon_loss"][-1] < 0.9 * logs["epoch_recon_loss"][0]
    assert logs["val_recon_loss"][-1] < 0.9 * logs["val_recon_loss"][0]
    assert m.is_trained_ and not m.model.training
    z = m.get_latent(mean=True)
    assert z.shape == (len(ad), 8) and z.dtype == np.float32 and np.isfinite(z).all()
    zm, zv = m.get_latent(ad.X, ad.obs["study"], mean_var=True)
    assert np.allclose(zm, z) and (zv > 0).all()
    zs = m.get_latent()                                    # sampled: differs from the mean by ~sqrt(var)
    assert zs.shape == z.shape and not np.allclose(zs, z)
    y = m.get_y()
    assert y.shape == (len(ad), 32) and (y >= 0).all()
    with pytest.raises(ValueError, match="Incorrect conditions"):
        m.get_latent(ad.X, np.array(["nope"] * len(ad)))
    # cell types separate in the latent space
    y_ct = ad.obs["cell_type"].to_numpy()
    idx = np.random.default_rng(0).permutation(len(ad))
    a, b = idx[: len(ad) // 2], idx[len(ad) // 2:]
    assert knn_acc(z[a], y_ct[a], z[b], y_ct[b]) > 0.9


def test_knn_accuracy_matches_oracle_training(trained):
    """North-star acceptance: post-training latent kNN label-transfer accuracy within 0.5 points of the same model
    trained by the CPU oracle (same init, data, schedule; independent minibatch / noise draws).  On this small
    problem both reach the ceiling, so the 0.5-point bar is meaningful and stable."""
    from oracle import cvae_oracle as O
    m, ad, adq, init = trained
    torch.set_num_threads(4)
    cfg = dict(G=ad.n_vars, C=3, H=[64, 32], Z=8, recon="nb", use_bn=False, use_ln=True, dr_rate=0.0, use_mmd=True,
               mmd_on="z", boundary=None, beta=1.0)
    X = torch.tensor(ad.X.toarray())
    enc = m.model.condition_encoder
    cond = np.array([enc[s] for s in ad.obs["study"]])
    P = {k: v.cpu().clone() for k, v in init.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    tr_idx = m.trainer.train_data                         # same split sizes; oracle uses its own sampler draws
    n, g, t = len(ad), torch.Generator().manual_seed(1), 0
    counts = np.bincount(cond)
    w = torch.tensor(1.0 / (n * counts[cond]))
    for epoch in range(30):
        ids = torch.multinomial(w, len(tr_idx), replacement=True, generator=g)
        alpha = min(0.5 * epoch / 20, 0.5)
        for lo in range(0, len(ids), 128):
            b = ids[lo:lo + 128].numpy()
            xb = X[b]
            eps = torch.randn(len(b), 8, generator=g)
            _, grads, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in P.items()}, cfg, xb, cond[b], xb.sum(1),
                                                 eps, alpha)
            t += 1
            for k in P:
                O.adam_step(P[k], grads[k], M[k], V[k], t, lr=1e-3, eps=0.01, wd=0.04)
    mu_o, _ = O.encoder(P, cfg, torch.log(1 + X), torch.tensor(cond), train=False)
    z_o = mu_o.numpy()
    z_h = m.get_latent(mean=True)
    y_ct = ad.obs["cell_type"].to_numpy()
    idx = np.random.default_rng(0).permutation(n)
    a, b = idx[: n // 2], idx[n // 2:]
    acc_o, acc_h = knn_acc(z_o[a], y_ct[a], z_o[b], y_ct[b]), knn_acc(z_h[a], y_ct[a], z_h[b], y_ct[b])
    assert abs(acc_o - acc_h) <= 0.005 + 1e-9, (acc_o, acc_h)


def test_knn_accuracy_paired_replay_of_reference_training_runs():
    """North-star acceptance, trajectory to trajectory: two training runs of the REAL reference (seeds 0 and 1; 2000
    cells x 128 genes, 10 epochs = 150 Adam steps; tests/golden/make_goldens_round2.py knn_paired) stored the sampler's
    minibatch ids (trainers/trvae/trainer.py:154-191), the reparameterisation noise and every dropout keep-mask of
    every iteration.  The HIP engine replays exactly those draws from the reference's initial weights (same loss
    composition / annealing / Adam as trainer.py:193-223,262-280): per seed the latent kNN label-transfer accuracy
    (utils/knn.py-style distance-weighted kNN on the encoder means) must agree within 0.5 points.
    Trajectory checks: over the first epoch (15 steps) the oracle - pinned to the reference, and within 3e-7 of the
    reference's final parameters when it replays all 150 steps on the CPU - runs in lockstep and the HIP parameters
    must stay within 2e-5 of it; at the end the parameters must be within 2e-2 of each tensor's scale and the latent
    means within 2e-2 (Frobenius).  The end bars are loose on purpose: two fp32 implementations with different summation
    orders cannot stay closer over 150 steps, because ReLU gates sit on knife edges - seed 0 has a normalised
    pre-activation of +-1e-7 at step 21 whose sign follows the order of earlier steps' float atomics (gradient of that
    step moves by 6e-3), seed 1 one of 2e-6 at step 111 (4e-2 of a late, small gradient); away from such steps the HIP
    trajectory tracks the oracle's to 8e-6 over all 150 steps (scripts/dbg/paired_traj.py prints both)."""
    import json
    import os
    from collections import OrderedDict
    from oracle import cvae_oracle as O
    from scarches_amd.engine import CVAEEngine
    from tests._util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "knn_paired.npz"))
    cfg = json.loads(str(z["cfg"]))
    n, G, C, Z, H, bs = cfg["n"], cfg["G"], cfg["C"], cfg["Z"], cfg["H"], cfg["batch_size"]
    X = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(n, G))
    cond, t, a, b = z["cond"], z["cell_type"], z["split_a"], z["split_b"]
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, dr_rate=cfg["dr_rate"], use_mmd=True,
                mmd_on="z", beta=1.0, seed=0)
    for seed in range(2):
        pre = f"s{seed}/"
        W0 = OrderedDict((k[len(pre) + 3:], torch.tensor(z[k])) for k in z.files if k.startswith(pre + "w0/"))
        W1 = {k[len(pre) + 3:]: z[k] for k in z.files if k.startswith(pre + "w1/")}
        eng = CVAEEngine(dict(spec), W0, "cuda:0")
        tr = z[pre + "train_idx"].astype(np.int64)
        Xt = X[tr].tocsr()
        Xt.sort_indices()
        eng.bind_csr(torch.tensor(Xt.indptr.astype(np.int64)).cuda(), torch.tensor(Xt.indices.astype(np.int32)).cuda(),
                     torch.tensor(Xt.data.astype(np.float32)).cuda())
        cond_t = torch.tensor(cond[tr].astype(np.int32)).cuda()
        samples = z[pre + "samples"].astype(np.int64)            # [epochs][n_train] rows of the training subset
        n_full = samples.shape[1] // bs
        tail = samples.shape[1] - n_full * bs
        keys = ("enc0", "enc1", "dec0", "dec1")
        widths = dict(enc0=H[0], enc1=H[1], dec0=H[1], dec1=H[0])
        i_full = i_tail = 0
        Po = {k: v.clone() for k, v in W0.items()}               # the oracle's replica for the lockstep epoch
        Mo, Vo = {k: torch.zeros_like(v) for k, v in Po.items()}, {k: torch.zeros_like(v) for k, v in Po.items()}
        Xd, ocfg, t_o = torch.tensor(Xt.toarray()), dict(spec, boundary=None), 0
        for epoch in range(cfg["epochs"]):
            alpha = min(cfg["alpha_kl"] * epoch / cfg["alpha_epoch_anneal"], cfg["alpha_kl"])
            for it in range(n_full + (1 if tail else 0)):
                rows = torch.tensor(samples[epoch, it * bs:(it + 1) * bs]).cuda()
                B = rows.numel()
                full = B == bs
                eps = torch.tensor(z[pre + ("eps_full" if full else "eps_tail")][i_full if full else i_tail]).cuda()
                masks = {}
                for k in keys:
                    bits = z[pre + ("mask_full/" if full else "mask_tail/") + k][i_full if full else i_tail]
                    masks[k] = torch.tensor(np.unpackbits(bits, axis=1)[:, :widths[k]].copy()).cuda().contiguous()
                if full:
                    i_full += 1
                else:
                    i_tail += 1
                eng.begin_step(alpha_kl=alpha, lr=cfg["lr"], eps=cfg["eps"])
                eng.forward_backward(rows, cond_t[rows].contiguous(), B, train=True, compute_grad=True, eps=eps,
                                     masks=masks)
                eng.adam_step(weight_decay=cfg["weight_decay"])
                if epoch == 0:
                    r = rows.cpu().numpy()
                    xb = Xd[r]
                    _, g_o, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in Po.items()}, ocfg, xb, cond[tr][r],
                                                       xb.sum(1), eps.cpu(), alpha, {k: v.cpu() for k, v in masks.items()})
                    t_o += 1
                    for k in Po:
                        O.adam_step(Po[k], g_o[k], Mo[k], Vo[k], t_o, lr=cfg["lr"], eps=cfg["eps"],
                                    wd=cfg["weight_decay"])
            if epoch == 0:
                for k in Po:
                    err = float((eng.p[k].cpu() - Po[k]).abs().max() / Po[k].abs().max())
                    assert err <= 2e-5, (seed, k, err)
        torch.cuda.synchronize()
        assert i_full == z[pre + "eps_full"].shape[0] and i_tail == z[pre + "eps_tail"].shape[0]
        for k, ref in W1.items():
            got = eng.p[k].detach().cpu().numpy()
            assert np.max(np.abs(got - ref)) <= 2e-2 * np.max(np.abs(ref)), (seed, k, np.max(np.abs(got - ref)))
        Xa = X.tocsr()
        Xa.sort_indices()
        eng.bind_csr(torch.tensor(Xa.indptr.astype(np.int64)).cuda(), torch.tensor(Xa.indices.astype(np.int32)).cuda(),
                     torch.tensor(Xa.data.astype(np.float32)).cuda())
        _, mu, _ = eng.encode(torch.arange(n, device="cuda"), torch.tensor(cond.astype(np.int32)).cuda(), n, mean=True)
        zl = mu.cpu().numpy()
        ref_lat = z[pre + "latent_mean"]
        assert np.linalg.norm(zl - ref_lat) <= 2e-2 * np.linalg.norm(ref_lat), np.linalg.norm(zl - ref_lat)
        acc, acc_ref = knn_acc(zl[a], t[a], zl[b], t[b]), float(z[pre + "acc"])
        print("paired kNN acceptance: seed", seed, "reference", round(acc_ref, 4), "hip", round(acc, 4))
        assert 0.7 < acc_ref < 0.95                               # below the ceiling: the bar can fail
        assert abs(acc - acc_ref) <= 0.005 + 1e-9, (seed, acc, acc_ref)


def test_knn_accuracy_matches_reference_training_below_the_ceiling():
    """The same acceptance on a HARD set (cell types that share 80 % of their profile; accuracy 0.86 - 0.99 from seed to
    seed): five models trained by the REAL reference (tests/golden/make_goldens_round2.py knn; seeds 0-4, the schedule
    in the fixture's cfg) stored their latent kNN accuracies, five HIP-trained models with the same seeds / schedule
    are compared with them.  Minibatch and noise draws differ between the two (the reference draws on the CPU
    generator), so the comparison is between means: |mean_hip - mean_ref| <= 0.5 points + 2 standard errors."""
    import json
    import os
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import TRVAE
    from tests._util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "knn_acceptance.npz"))
    cfg = json.loads(str(z["cfg"]))
    X = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(cfg["n"], cfg["G"]))
    c, t, a, b = z["cond"], z["cell_type"], z["split_a"], z["split_b"]
    names = np.array([f"s{i}" for i in range(cfg["C"])])
    ad = CellData(X, pd.DataFrame({"study": names[c], "cell_type": [f"t{i}" for i in t]}))
    ref = z["ref_acc"]
    assert 0.80 <= ref.mean() <= 0.95 and ref.min() < 0.9          # below the ceiling, by construction
    accs = []
    for seed in range(len(ref)):
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(seed)
            m = TRVAE(ad, condition_key="study", conditions=list(names), hidden_layer_sizes=cfg["H"],
                      latent_dim=cfg["Z"], dr_rate=cfg["dr_rate"], use_mmd=True, recon_loss="nb")
            m.train(n_epochs=cfg["epochs"], lr=1e-3, eps=0.01, batch_size=cfg["batch_size"],
                    alpha_epoch_anneal=cfg["alpha_epoch_anneal"], alpha_kl=cfg["alpha_kl"], seed=seed,
                    early_stopping_kwargs=dict(patience=50), monitor=False)
        zl = m.get_latent(ad.X, ad.obs["study"], mean=True)
        accs.append(knn_acc(zl[a], t[a], zl[b], t[b]))
    accs = np.array(accs)
    se = np.sqrt(ref.var(ddof=1) / len(ref) + accs.var(ddof=1) / len(accs))
    print("kNN acceptance: reference", ref.round(4), "hip", accs.round(4), "se", round(float(se), 4))
    assert abs(accs.mean() - ref.mean()) <= 0.005 + 2 * se, (ref, accs)
    assert accs.min() > ref.min() - 0.08 and accs.max() <= 1.0


def test_surgery_save_load_and_query_training(trained, tmp_path):
    from scarches_amd.models import TRVAE
    m, ad, adq, init = trained
    d = str(tmp_path / "ref_model")
    m.save(d, overwrite=True)
    with contextlib.redirect_stdout(io.StringIO()):
        q = TRVAE.load_query_data(adq, d)
    assert q.conditions_[:3] == m.conditions_ and set(q.conditions_[3:]) == {"s3", "s4"}
    trainable = [n for n, p in q.model.named_parameters() if p.requires_grad]
    assert trainable == ["theta", "encoder.FC.L0.cond_L.weight", "decoder.FirstL.L0.cond_L.weight"]
    before = {k: v.detach().cpu().clone() for k, v in q.model.state_dict().items()}
    with contextlib.redirect_stdout(io.StringIO()):
        q.train(n_epochs=15, batch_size=128, alpha_epoch_anneal=10, weight_decay=0, monitor=False,
                early_stopping_kwargs=dict(patience=50))
    after = {k: v.detach().cpu() for k, v in q.model.state_dict().items()}
    C_old = len(m.conditions_)
    for k in after:
        if k in trainable:
            assert not torch.equal(before[k], after[k]), k
            # old condition columns: no query cell carries them and weight_decay=0 -> untouched
            assert torch.equal(before[k][..., :C_old], after[k][..., :C_old]), k
        else:
            assert torch.equal(before[k], after[k]), k          # frozen weights did not move
    assert q.trainer.logs["val_recon_loss"][-1] < q.trainer.logs["val_recon_loss"][0]
    # label transfer reference -> query in the joint latent space
    zr, zq = m.get_latent(mean=True), q.get_latent(mean=True)
    acc = knn_acc(zr, ad.obs["cell_type"].to_numpy(), zq, adq.obs["cell_type"].to_numpy())
    assert acc > 0.85, acc
    # the query model also maps reference cells exactly like the reference model (old weights/columns unchanged)
    zr2 = q.get_latent(ad.X, ad.obs["study"], mean=True)
    assert np.allclose(zr, zr2, atol=1e-6)


def test_module_forward_api():
    """trVAE.forward(x, batch, sizefactor) on dense device tensors returns the three 0-d losses."""
    from scarches_amd.models import trVAE
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        net = trVAE(50, ["a", "b"], [32, 16], 5, 0.0, use_mmd=True).cuda()
    x = torch.poisson(torch.rand(12, 50) * 2).cuda()
    b = torch.tensor([0, 1] * 6).cuda()
    net.eval()
    r, k, mm = net(x=x, batch=b, sizefactor=x.sum(1))
    assert r.dim() == 0 and torch.isfinite(r) and torch.isfinite(k) and torch.isfinite(mm) and r.item() > 0
    z = net.get_latent(x, b, mean=True)
    assert z.shape == (12, 5)
    with pytest.raises(Exception):
        trVAE(50, ["a"], [32], 5, recon_loss="poisson")


def test_batchnorm_model_trains_and_freezes_running_stats(tmp_path):
    """use_bn=True: running statistics move during reference training, are saved / loaded with the state dict, and
    stay put during frozen query training (trainers/trvae/trainer.py:264-270)."""
    from scarches_amd.anndata_lite import CellData
    from scarches_amd.models import TRVAE
    X, obs = synth(900, 96, 3, 4, seed=11)
    is_q = (obs["study"] == "s2").to_numpy()
    ad, adq = CellData(X[~is_q], obs[~is_q]), CellData(X[is_q], obs[is_q])
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(0)
        m = TRVAE(ad, condition_key="study", hidden_layer_sizes=[32], latent_dim=6, use_bn=True, use_ln=False,
                  recon_loss="zinb", use_mmd=False)
        m.train(n_epochs=8, batch_size=128, alpha_epoch_anneal=5, use_early_stopping=False, monitor=False)
    sd = m.model.state_dict()
    assert float(sd["encoder.FC.N0.running_mean"].abs().max()) > 0
    assert int(sd["encoder.FC.N0.num_batches_tracked"]) > 0
    logs = m.trainer.logs
    assert logs["epoch_recon_loss"][-1] < logs["epoch_recon_loss"][0]
    lat = m.get_latent(mean=True)
    assert np.isfinite(lat).all()
    m.save(str(tmp_path / "bn"), overwrite=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m2 = TRVAE.load(str(tmp_path / "bn"), ad)
        assert np.allclose(m2.get_latent(mean=True), lat, rtol=1e-5, atol=1e-6)
        q = TRVAE.load_query_data(adq, m)
        rm0 = q.model.state_dict()["encoder.FC.N0.running_mean"].cpu().clone()
        q.train(n_epochs=3, batch_size=128, use_early_stopping=False, monitor=False, weight_decay=0)
    assert torch.equal(q.model.state_dict()["encoder.FC.N0.running_mean"].cpu(), rm0)
    assert np.isfinite(q.get_latent(mean=True)).all()


def test_adaptors_rebuild_query_model_on_device(trained, tmp_path):
    """`Adaptor` + `attach_adaptors` (models/trvae/adaptors.py) with parameters living in the engine's flat HBM
    buffer: the adaptors of the two query studies, attached to the reference model, give a model that maps the query
    cells exactly like the surgery-trained query model (frozen surgery only moves the adaptor's tensors)."""
    from scarches_amd.models import TRVAE, Adaptor, attach_adaptors
    m, ad, adq, init = trained
    with contextlib.redirect_stdout(io.StringIO()):
        q = TRVAE.load_query_data(adq, m)
        q.train(n_epochs=3, batch_size=128, alpha_epoch_anneal=10, weight_decay=0, monitor=False,
                early_stopping_kwargs=dict(patience=50))
    zq = q.get_latent(mean=True)
    new = [c for c in q.conditions_ if c not in m.conditions_]
    paths = []
    for c in new:
        a = Adaptor(q, c)
        assert all(v.shape[1] == 1 for v in a.cond_params.values())
        paths.append(str(tmp_path / f"adaptor_{c}"))
        a.save(paths[-1])
    zr = m.get_latent(mean=True)
    with contextlib.redirect_stdout(io.StringIO()):
        big = attach_adaptors(m, [paths[0], Adaptor(q, new[1])])
    assert big.conditions_ == m.conditions_ + new and big.model.n_conditions == len(m.conditions_) + 2
    z_big = big.get_latent(adq.X, adq.obs["study"], mean=True)
    assert np.allclose(z_big, zq, rtol=1e-6, atol=1e-6)
    assert np.allclose(big.get_latent(ad.X, ad.obs["study"], mean=True), zr, atol=1e-6)
    assert np.allclose(m.get_latent(mean=True), zr, atol=0)           # the passed model still works, unchanged
