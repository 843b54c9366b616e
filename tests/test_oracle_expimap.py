"""CPU suite: the expiMap oracle (oracle/expimap_oracle.py) against fixtures produced by the REAL reference
(tests/golden/make_goldens_expimap.py).  This is what pins that oracle."""
import numpy as np
import pytest
import torch

from oracle import cvae_oracle as O
from oracle import expimap_oracle as E
from tests._util import GOLDEN, dense_from_case, golden_ref64, grads_close, load_case, rel_err

torch.set_num_threads(1)
TOL = 1e-5
EXPIMAP_CASES = ["expimap_nb_hard", "expimap_nb_ext", "expimap_mse_soft", "expimap_nb_hsic_ova",
                 # decoder last layers other than softmax / identity and the library-size encoder
                 "expimap_nb_softplus", "expimap_nb_exp", "expimap_nb_relu", "expimap_mse_relu", "expimap_mse_softplus",
                 "expimap_nb_lenc"]


def _eps_l(z):
    return torch.tensor(z["eps_l"]) if "eps_l" in z.files else None


def _params(W):
    return {k: v for k, v in W.items() if not k.endswith(".mask")}


def _masks(z, cfg):
    mask = torch.tensor(z["mask"])
    ext_mask = torch.tensor(z["ext_mask"]) if "ext_mask" in z.files else None
    return mask, ext_mask


@pytest.mark.parametrize("name", EXPIMAP_CASES)
def test_expimap_losses_and_grads(name):
    z, cfg, W, Gr, masks = load_case(name)
    x = dense_from_case(z, cfg["G"])
    mask, ext_mask = _masks(z, cfg)
    res, grads, aux = E.expimap_loss_and_grads(_params(W), cfg, x, z["batch"], torch.tensor(z["sizefactor"]),
                                               torch.tensor(z["eps"]), float(z["alpha"]), float(z["beta"]), mask,
                                               ext_mask, masks or None, eps_l=_eps_l(z))
    assert abs(res["recon"] - float(z["recon"])) <= TOL * abs(float(z["recon"]))
    assert abs(res["kl"] - float(z["kl"])) <= TOL * abs(float(z["kl"]))
    if cfg["use_hsic"]:
        # sqrt of a difference of O(1) means: fp32 cancellation noise of the reference itself
        assert abs(res["hsic"] - float(z["hsic"])) <= 1e-4 * abs(float(z["hsic"]))
    bad = grads_close({k: v.numpy() for k, v in grads.items()}, Gr, rtol=1e-5, ref64=golden_ref64(name)[1])
    assert not bad, bad
    # hard-masked entries carry exactly zero gradient
    if not cfg["soft_mask"]:
        g = grads["decoder.L0.expr_L.weight"].numpy()
        assert np.all(g[z["mask"].T == 0] == 0)


@pytest.mark.parametrize("name", ["expimap_nb_hard", "expimap_nb_ext"])
def test_expimap_latent_eval(name):
    z, cfg, W, _, _ = load_case(name)
    P = {k: torch.tensor(v) for k, v in _params(W).items()}
    x = dense_from_case(z, cfg["G"])
    mu, lv = E.ext_encoder(P, cfg, torch.log(1 + x), torch.tensor(z["batch"]), train=False)
    assert rel_err(mu.numpy(), z["z_mean"]) < TOL
    assert rel_err((torch.exp(lv) + 1e-4).numpy(), z["z_var"]) < TOL


@pytest.mark.parametrize("name", ["expimap_nb_hard", "expimap_nb_ext", "expimap_mse_soft", "expimap_nb_softplus",
                                  "expimap_nb_lenc"])
def test_expimap_adam_prox_steps(name):
    """Adam + proximal operators in the order of expiMapTrainer.on_iteration (regularized.py:283-298)."""
    z, cfg, W, _, masks = load_case(name)
    x = dense_from_case(z, cfg["G"])
    mask, ext_mask = _masks(z, cfg)
    steps, wd, lr = int(z["adam_steps"]), float(z["adam_wd"]), 1e-3
    P = {k: torch.tensor(v.copy()) for k, v in _params(W).items()}
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    omega = torch.tensor(z["omega"]) if "omega" in z.files else None
    for t in range(1, steps + 1):
        _, grads, _ = E.expimap_loss_and_grads({k: v.numpy() for k, v in P.items()}, cfg, x, z["batch"],
                                               torch.tensor(z["sizefactor"]), torch.tensor(z["eps"]),
                                               float(z["alpha"]), float(z["beta"]), mask, ext_mask, masks or None,
                                               eps_l=_eps_l(z))
        for k in P:
            O.adam_step(P[k], grads[k], M[k], V[k], t, lr=lr, eps=0.01, wd=wd)
        if cfg["alpha_l1"] is not None and cfg["soft_mask"]:
            E.prox_l1(P["decoder.L0.expr_L.weight"], cfg["alpha_l1"] * lr, mask.T)
        if cfg["alpha_gl"] is not None:
            E.prox_group_lasso(P["decoder.L0.expr_L.weight"], cfg["alpha_gl"] * lr, omega)
        if cfg["gamma_ext"] is not None and cfg["n_ext"] > 0:
            E.prox_l1(P["decoder.L0.ext_L.weight"], cfg["gamma_ext"] * lr)
        if cfg["alpha_l1"] is not None and cfg["n_ext_m"] > 0 and cfg["soft_ext_mask"]:
            E.prox_l1(P["decoder.L0.ext_L_m.weight"], cfg["alpha_l1"] * lr, ext_mask.T)
    for k in P:
        ref = z["adam/" + k]
        assert np.max(np.abs(P[k].numpy() - ref)) <= 3e-6 * max(1.0, np.max(np.abs(ref))), k


def test_prox_ops_and_hsic():
    z = np.load(GOLDEN + "/expimap_ops.npz")
    W, I, om, a = z["prox_W"], z["prox_I"], z["prox_omega"], float(z["prox_alpha"])
    assert np.array_equal(E.prox_group_lasso(torch.tensor(W.copy()), a).numpy(), z["prox_gl"])
    assert np.array_equal(E.prox_group_lasso(torch.tensor(W.copy()), a, torch.tensor(om)).numpy(),
                          z["prox_gl_omega"])
    assert np.array_equal(E.prox_l1(torch.tensor(W.copy()), a).numpy(), z["prox_l1"])
    assert np.array_equal(E.prox_l1(torch.tensor(W.copy()), a, torch.tensor(I)).numpy(), z["prox_l1_I"])
    for key in "abc":
        zz = torch.tensor(z[f"hsic_{key}_z"], requires_grad=True)
        ss = torch.tensor(z[f"hsic_{key}_s"], requires_grad=True)
        h = E.hsic(zz, ss)
        h.backward()
        assert abs(h.item() - float(z[f"hsic_{key}_val"])) <= 1e-5 * float(z[f"hsic_{key}_val"])
        assert rel_err(zz.grad.numpy(), z[f"hsic_{key}_dz"]) < 1e-4
        assert rel_err(ss.grad.numpy(), z[f"hsic_{key}_ds"]) < 1e-4
        h64 = E.hsic(zz.detach().double(), ss.detach().double())
        assert abs(h64.item() - float(z[f"hsic_{key}_val64"])) <= 1e-10
