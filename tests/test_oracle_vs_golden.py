"""CPU suite: the oracle (oracle/cvae_oracle.py) against fixtures produced by the REAL reference
(tests/golden/make_goldens.py).  This is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

from oracle import cvae_oracle as O
from tests._util import TRVAE_CASES, dense_from_case, golden_ref64, grads_close, load_case, rel_err, GOLDEN

torch.set_num_threads(1)
TOL = 1e-5          # north_star: losses and gradients within rel 1e-5 (fp32)


@pytest.mark.parametrize("name", TRVAE_CASES)
def test_trvae_losses_and_grads(name):
    z, cfg, W, Gr, masks = load_case(name)
    x = dense_from_case(z, cfg["G"])
    res, grads, aux = O.trvae_loss_and_grads(W, cfg, x, z["batch"], torch.tensor(z["sizefactor"]),
                                             torch.tensor(z["eps"]), float(z["alpha"]), masks or None)
    assert abs(res["recon"] - float(z["recon"])) <= TOL * abs(float(z["recon"]))
    assert abs(res["kl"] - float(z["kl"])) <= TOL * abs(float(z["kl"]))
    if cfg["use_mmd"]:
        # the reference sums C(C-1)/2 differences of O(10) block means in fp32; allow its own fp32 noise
        assert abs(res["mmd"] - float(z["mmd"])) <= 2e-5 * abs(float(z["mmd"])) + 1e-5
    assert not grads_close({k: v.numpy() for k, v in grads.items()}, Gr, rtol=1e-5, ref64=golden_ref64(name)[1])
    # f64 yardstick: the oracle in f64 agrees with the reference in f64 much tighter
    res64, _, _ = O.trvae_loss_and_grads(W, cfg, x, z["batch"], torch.tensor(z["sizefactor"]),
                                         torch.tensor(z["eps"]), float(z["alpha"]), masks or None,
                                         dtype=torch.float64)
    assert abs(res64["recon"] - float(z["recon64"])) <= 1e-9 * abs(float(z["recon64"]))
    assert abs(res64["kl"] - float(z["kl64"])) <= 1e-9 * abs(float(z["kl64"]))


@pytest.mark.parametrize("name", ["trvae_nb_ln_small", "trvae_zinb_bn"])
def test_latent_eval(name):
    z, cfg, W, _, _ = load_case(name)
    P = {k: torch.tensor(v) for k, v in W.items()}
    x = dense_from_case(z, cfg["G"])
    x_in = x if cfg["recon"] == "mse" else torch.log(1 + x)
    if cfg["use_bn"]:
        # the reference ran one train-mode forward first: BatchNorm running stats moved (momentum 0.1)
        cfg["_bn_out"] = {}
        O.trvae_forward(P, cfg, x, torch.tensor(z["batch"]), torch.tensor(z["sizefactor"]),
                        torch.tensor(z["eps"]), train=True)
        for k, v in cfg.pop("_bn_out").items():
            assert rel_err(v.numpy(), z["bn_after/" + k]) < TOL, k
            P[k] = v
    mu, logvar = O.encoder(P, cfg, x_in, torch.tensor(z["batch"]), train=False)
    assert rel_err(mu.numpy(), z["z_mean"]) < TOL
    assert rel_err((torch.exp(logvar) + 1e-4).numpy(), z["z_var"]) < TOL


@pytest.mark.parametrize("name", ["trvae_nb_ln_small", "trvae_nb_ragged"])
def test_adam_steps(name):
    z, cfg, W, _, _ = load_case(name)
    x = dense_from_case(z, cfg["G"])
    steps, wd = int(z["adam_steps"]), float(z["adam_wd"])
    P = {k: torch.tensor(v.copy()) for k, v in W.items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    for t in range(1, steps + 1):
        _, grads, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in P.items()}, cfg, x, z["batch"],
                                             torch.tensor(z["sizefactor"]), torch.tensor(z["eps"]),
                                             float(z["alpha"]))
        for k in P:
            O.adam_step(P[k], grads[k], M[k], V[k], t, lr=1e-3, eps=0.01, wd=wd)
    for k in P:
        ref = z["adam/" + k]
        assert np.max(np.abs(P[k].numpy() - ref)) <= 2e-6 * max(1.0, np.max(np.abs(ref))), k


def test_nb_zinb_elementwise():
    z = np.load(GOLDEN + "/ops.npz")
    x, mu, th, pi = (torch.tensor(z[k]) for k in ("nb_x", "nb_mu", "nb_theta", "zinb_pi"))
    mu.requires_grad_(True)
    th.requires_grad_(True)
    pi.requires_grad_(True)
    ll = O.nb_loglik(x, mu, th)
    assert rel_err(ll.detach().numpy(), z["nb_ll"]) < 1e-6
    ll.sum().backward()
    assert rel_err(mu.grad.numpy(), z["nb_dmu"]) < 1e-5
    assert rel_err(th.grad.numpy(), z["nb_dtheta"]) < 1e-5
    mu.grad = th.grad = None
    zl = O.zinb_loglik(x, mu, th, pi)
    assert rel_err(zl.detach().numpy(), z["zinb_ll"]) < 1e-6
    zl.sum().backward()
    assert rel_err(pi.grad.numpy(), z["zinb_dpi"]) < 1e-5
    assert rel_err(mu.grad.numpy(), z["zinb_dmu"]) < 1e-5


@pytest.mark.parametrize("key", list("abcde"))
def test_mmd_block_identity(key):
    z = np.load(GOLDEN + "/ops.npz")
    c = json.loads(str(z[f"mmd_{key}_cfg"]))
    y = torch.tensor(z[f"mmd_{key}_y"], requires_grad=True)
    lab = torch.tensor(z[f"mmd_{key}_c"])
    val = O.mmd_loss(y, lab, c["C"], c["beta"], c["boundary"])
    ref = float(z[f"mmd_{key}_val"])
    # f64 evaluation of the same identity is the yardstick for the reference's own fp32 noise
    val64 = O.mmd_loss(y.detach().double(), lab, c["C"], c["beta"], c["boundary"]).item()
    noise = abs(ref - val64)
    assert abs(val.item() - ref) <= 1e-5 * abs(ref) + 2 * noise + 1e-6
    if val.requires_grad and ref != 0:
        val.backward()
        assert rel_err(y.grad.numpy(), z[f"mmd_{key}_grad"]) < 5e-5


def test_early_stopping_sequences():
    z = np.load(GOLDEN + "/host.npz")
    for seq_k, rec_k, kw in (("es_seq", "es_rec", {}),
                             ("es2_seq", "es2_rec", dict(patience=5, lr_patience=2, threshold=0.05, lr_factor=0.5))):
        es = O.EarlyStoppingOracle(**kw)
        rec = []
        for v in z[seq_k]:
            upd = es.update_state(v)
            cont, lr = es.step(v)
            rec.append((int(upd), int(cont), int(lr)))
            if not cont:
                break
        assert np.array_equal(np.array(rec, dtype=np.int32), z[rec_k])


def test_surgery_expand_and_freeze():
    z = np.load(GOLDEN + "/surgery.npz")
    old = {k[6:]: z[k] for k in z.files if k.startswith("ref_w/")}
    q = {k[4:]: z[k] for k in z.files if k.startswith("q_w/")}
    # the fresh query model's new columns are whatever the reference drew; take them from q itself
    exp = O.expand_state_dict(old, q)
    for k in q:
        assert np.array_equal(exp[k], q[k]), k
        if old[k].shape != q[k].shape:
            assert np.array_equal(q[k][..., :old[k].shape[-1]], old[k]), k
    assert O.surgery_trainable(list(q.keys())) == list(z["q_trainable"])
    # query forward/backward with frozen weights
    cfg = json.loads(str(z["cfg"]))
    cfg.update(C=len(z["q_conditions"]), recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z",
               boundary=None, beta=1.0, dr_rate=0.0)
    import scipy.sparse as sp
    B = len(z["q_batch"])
    x = torch.tensor(sp.csr_matrix((z["q_x_data"], z["q_x_indices"], z["q_x_indptr"]),
                                   shape=(B, cfg["G"])).toarray().astype(np.float32))
    res, grads, aux = O.trvae_loss_and_grads(q, cfg, x, z["q_batch"], x.sum(1), torch.tensor(z["q_eps"]),
                                             float(z["q_alpha"]), trainable=list(z["q_trainable"]))
    assert abs(res["recon"] - float(z["q_recon"])) <= TOL * abs(float(z["q_recon"]))
    assert abs(res["kl"] - float(z["q_kl"])) <= TOL * abs(float(z["q_kl"]))
    assert abs(res["mmd"] - float(z["q_mmd"])) <= 2e-5 * abs(float(z["q_mmd"])) + 1e-5
    for k in z["q_trainable"]:
        assert rel_err(grads[str(k)].numpy(), z["q_g/" + str(k)]) < 5e-5, k
    # eval-mode latent means through the wrapper
    P = {k: torch.tensor(v) for k, v in q.items()}
    mu, _ = O.encoder(P, cfg, torch.log(1 + x), torch.tensor(z["q_batch"]), train=False)
    assert rel_err(mu.numpy(), z["q_latent_mean"]) < TOL
