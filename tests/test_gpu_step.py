"""GPU parity of the whole HIP train step against the REAL reference's outputs (tests/golden/*.npz) and the oracle:
per-batch losses and every parameter gradient within rel 1e-5-class tolerances (stated per assertion), Adam updates."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests._util import TRVAE_CASES, golden_ref64, grads_close, load_case, rel_err

pytestmark = pytest.mark.gpu

LN_CASES = TRVAE_CASES            # includes the BatchNorm case trvae_zinb_bn
BUFFERS = ("running_mean", "running_var", "num_batches_tracked")


def make_engine(name):
    from scarches_amd.engine import CVAEEngine
    z, cfg, W, Gr, masks = load_case(name)
    spec = dict(cfg)
    params = OrderedDict((k, torch.tensor(v)) for k, v in W.items() if not k.endswith(BUFFERS))
    spec["buffers"] = {k: torch.tensor(v) for k, v in W.items() if k.endswith(BUFFERS)}
    eng = CVAEEngine(spec, params, "cuda")
    eng.bind_csr(torch.tensor(z["x_indptr"]).cuda(), torch.tensor(z["x_indices"]).cuda(),
                 torch.tensor(z["x_data"]).cuda())
    return eng, z, cfg, W, Gr, masks


def run_step(eng, z, cfg, masks, alpha):
    B = cfg["B"]
    cond = torch.tensor(z["batch"]).to(torch.int32).cuda()
    eps = torch.tensor(z["eps"]).cuda()
    dmasks = {k: v.to(torch.uint8).cuda().contiguous() for k, v in masks.items()}
    eng.set_hyper(alpha_kl=alpha)
    eng.forward_backward(None, cond, B, train=True, compute_grad=True, eps=eps, masks=dmasks)
    torch.cuda.synchronize()
    return eng.losses.cpu().numpy()


@pytest.mark.parametrize("name", LN_CASES)
def test_step_losses_and_grads(name):
    from oracle import cvae_oracle as O
    eng, z, cfg, W, Gr, masks = make_engine(name)
    alpha = float(z["alpha"])
    losses = run_step(eng, z, cfg, masks, alpha)
    recon, kl, mmd = (float(z[k]) for k in ("recon", "kl", "mmd"))
    # north_star tolerance: rel 1e-5 (fp32) on per-batch losses
    assert abs(losses[0] - recon) <= 1e-5 * abs(recon), (losses[0], recon)
    assert abs(losses[1] - kl) <= 1e-5 * abs(kl), (losses[1], kl)
    if cfg["use_mmd"]:
        # MMD is a sum of C(C-1)/2 differences of O(10) block means: the reference's own fp32 evaluation carries
        # cancellation noise, measured here against the f64 oracle and added to the rel-1e-5 budget
        res64 = golden_ref64(name)[0]
        noise = abs(mmd - res64["mmd"])
        assert abs(losses[2] - mmd) <= 1e-5 * abs(mmd) + 2 * noise + 1e-6, (losses[2], mmd, res64["mmd"])
    assert abs(losses[3] - (recon + alpha * kl + mmd)) <= 2e-5 * abs(recon + alpha * kl + mmd)
    got = {k: eng.g[k].cpu().numpy() for k in eng.names}
    # north_star tolerance: rel 1e-5 per gradient tensor, plus twice the reference's own fp32 distance from the fp64
    # yardstick (measured on this case; tests/_util.py grads_close)
    bad = grads_close(got, Gr, rtol=1e-5, floor_frac=1e-7, ref64=golden_ref64(name)[1], tag=name)
    assert not bad, bad


@pytest.mark.parametrize("name", ["trvae_nb_ln_small", "trvae_nb_ragged"])
def test_adam_steps_match_reference(name):
    eng, z, cfg, W, Gr, masks = make_engine(name)
    steps, wd = int(z["adam_steps"]), float(z["adam_wd"])
    B = cfg["B"]
    cond = torch.tensor(z["batch"]).to(torch.int32).cuda()
    eps = torch.tensor(z["eps"]).cuda()
    for _ in range(steps):
        eng.begin_step(alpha_kl=float(z["alpha"]), lr=1e-3, eps=0.01, clip=0.0)
        eng.forward_backward(None, cond, B, train=True, compute_grad=True, eps=eps)
        eng.adam_step(weight_decay=wd)
    torch.cuda.synchronize()
    for k in eng.names:
        ref = z["adam/" + k]
        err = np.max(np.abs(eng.p[k].cpu().numpy() - ref))
        assert err <= 3e-6 * max(1.0, np.max(np.abs(ref))), (k, err)


def test_batchnorm_running_stats_and_eval_latent():
    """BatchNorm: one train-mode forward moves the running statistics like the reference's; the eval-mode latent
    (running statistics) then matches the reference's get_latent."""
    eng, z, cfg, W, Gr, masks = make_engine("trvae_zinb_bn")
    run_step(eng, z, cfg, masks, float(z["alpha"]))
    for k in [f for f in z.files if f.startswith("bn_after/")]:
        name = k[len("bn_after/"):]
        assert rel_err(eng.buffers[name].cpu().numpy(), z[k]) < 1e-5, name
    assert int(eng.buffers["encoder.FC.N0.num_batches_tracked"]) == 1
    cond = torch.tensor(z["batch"]).to(torch.int32).cuda()
    zz, mu, var = eng.encode(None, cond, cfg["B"], mean=True)
    torch.cuda.synchronize()
    assert rel_err(mu.cpu().numpy(), z["z_mean"]) < 1e-5
    assert rel_err(var.cpu().numpy(), z["z_var"]) < 1e-5


def test_eval_latent_matches_reference():
    eng, z, cfg, W, Gr, masks = make_engine("trvae_nb_ln_small")
    B = cfg["B"]
    cond = torch.tensor(z["batch"]).to(torch.int32).cuda()
    zz, mu, var = eng.encode(None, cond, B, mean=True)
    torch.cuda.synchronize()
    assert rel_err(mu.cpu().numpy(), z["z_mean"]) < 1e-5
    assert rel_err(var.cpu().numpy(), z["z_var"]) < 1e-5
    assert rel_err(zz.cpu().numpy(), z["z_mean"]) < 1e-5


def test_frozen_surgery_step():
    """Query step with the reference's freeze set: only theta / cond_L weights receive gradients and move."""
    import json
    from scarches_amd.engine import CVAEEngine
    from tests._util import GOLDEN
    z = np.load(GOLDEN + "/surgery.npz")
    cfg = json.loads(str(z["cfg"]))
    q = OrderedDict((k[4:], torch.tensor(z[k])) for k in z.files if k.startswith("q_w/"))
    spec = dict(G=cfg["G"], C=len(z["q_conditions"]), H=cfg["H"], Z=cfg["Z"], recon="nb", use_bn=False, use_ln=True,
                use_mmd=True, mmd_on="z", boundary=None, beta=1.0, dr_rate=0.0)
    eng = CVAEEngine(spec, q, "cuda")
    eng.set_trainable([str(s) for s in z["q_trainable"]])
    eng.bind_csr(torch.tensor(z["q_x_indptr"]).cuda(), torch.tensor(z["q_x_indices"]).cuda(),
                 torch.tensor(z["q_x_data"]).cuda())
    B = len(z["q_batch"])
    eng.set_hyper(alpha_kl=float(z["q_alpha"]))
    eng.forward_backward(None, torch.tensor(z["q_batch"]).to(torch.int32).cuda(), B, eps=torch.tensor(z["q_eps"]).cuda())
    torch.cuda.synchronize()
    L = eng.losses.cpu().numpy()
    assert abs(L[0] - float(z["q_recon"])) <= 1e-5 * abs(float(z["q_recon"]))
    assert abs(L[1] - float(z["q_kl"])) <= 1e-5 * abs(float(z["q_kl"]))
    assert abs(L[2] - float(z["q_mmd"])) <= 5e-5 * abs(float(z["q_mmd"])) + 1e-5
    ref = {str(k): z["q_g/" + str(k)] for k in z["q_trainable"]}
    got = {k: eng.g[k].cpu().numpy() for k in ref}
    assert not grads_close(got, ref, rtol=1e-5, floor_frac=1e-7, tag="surgery")
    for k in eng.names:           # frozen tensors: no gradient written
        if k not in ref:
            assert float(eng.g[k].abs().max()) == 0.0, k
    before = {k: eng.p[k].clone() for k in eng.names}
    eng.begin_step(alpha_kl=0.5, lr=1e-3)
    eng.adam_step(weight_decay=0.0)
    torch.cuda.synchronize()
    for k in eng.names:
        moved = not torch.equal(before[k], eng.p[k])
        assert moved == (k in ref), k
