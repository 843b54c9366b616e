"""Generate golden fixtures by RUNNING THE REAL REFERENCE (build container only).

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

Every array written here is data (inputs + the reference's outputs); no
reference source text is stored.  The GPU box never runs this script - the
`.npz` files are committed and travel with the repo.

Reference entry points exercised (paths under /root/reference/scarches):
  models/trvae/trvae.py:116-157      trVAE.forward
  models/trvae/losses.py:63-270      nb / zinb / mmd
  models/base/_base.py:199-281       SurgeryMixin.load_query_data
  models/base/_base.py:285-340       CVAELatentsMixin.get_latent
  trainers/trvae/trainer.py:211-325  Adam hyper-parameters, early stopping
  trainers/trvae/_utils.py:73-143    train_test_split
  dataset/trvae/anndata.py:12-123    AnnotatedDataset
  utils/monitor.py:4-115             EarlyStopping
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
from _ref_import import FakeAnnData  # noqa: E402

torch.set_num_threads(1)

from scarches.models.trvae.trvae import trVAE  # noqa: E402
from scarches.models.trvae.trvae_model import TRVAE  # noqa: E402
from scarches.models.trvae import losses as ref_losses  # noqa: E402
from scarches.trainers.trvae._utils import train_test_split, custom_collate  # noqa: E402
from scarches.dataset.trvae.anndata import AnnotatedDataset  # noqa: E402
from scarches.dataset.trvae._utils import label_encoder  # noqa: E402
from scarches.utils.monitor import EarlyStopping  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def synth_counts(rng, n, g, n_cond, density=0.12, lib=300.0):
    """Synthetic NB-ish counts, CSR, with unequal condition sizes."""
    prof = rng.gamma(0.3, 1.0, size=(4, g))
    prof /= prof.sum(1, keepdims=True)
    t = rng.integers(0, 4, size=n)
    s = lib * np.exp(rng.normal(0, 0.4, size=n))
    mean = s[:, None] * prof[t]
    theta = rng.gamma(2.0, 1.0, size=g) + 0.1
    lam = rng.gamma(theta, mean / theta)
    x = rng.poisson(lam).astype(np.float32)
    keep = rng.random(x.shape) < (density * x.shape[1] * 2 / max(1.0, (x > 0).sum(1).mean()))
    x = x * keep
    p = rng.dirichlet(np.full(n_cond, 2.0))
    cond = rng.choice(n_cond, size=n, p=p).astype(np.int64)
    return sp.csr_matrix(x), cond


class FixedDropout(torch.nn.Module):
    """nn.Dropout with an injected keep-mask: y = x * m / (1 - p)."""

    def __init__(self, p, mask):
        super().__init__()
        self.p, self.mask = p, mask

    def forward(self, x):
        return x * self.mask.to(x.dtype) / (1.0 - self.p)


def run_trvae_case(name, rng, G, C, H, Z, B, recon, use_bn, use_ln, use_mmd, mmd_on, boundary, beta,
                   dr_rate=0.0, cond_override=None, zero_row=False, adam_steps=0, wd=0.04, seed=0):
    torch.manual_seed(seed)
    conds = [f"c{i}" for i in range(C)]
    model = quiet(trVAE, G, conds, list(H), Z, dr_rate, use_mmd, mmd_on, boundary, recon, beta, use_bn, use_ln)
    model.train()
    xs, cond = synth_counts(rng, B, G, C)
    if cond_override is not None:
        cond = np.asarray(cond_override, dtype=np.int64)
    if zero_row:
        xd = xs.toarray()
        xd[1, :] = 0
        xs = sp.csr_matrix(xd)
    xs.sort_indices()
    x = torch.tensor(xs.toarray())
    if recon == 'mse':
        # mse models are fed normalised log data in the reference; keep it sparse-friendly
        x = torch.log1p(x)
        xs = sp.csr_matrix(x.numpy())
    sizefactor = torch.tensor(np.ravel(xs.sum(1)).astype(np.float32))
    batch = torch.tensor(cond)
    eps = torch.randn(B, Z)
    model.sampling = lambda mu, lv: mu + eps.to(mu.dtype) * torch.sqrt(torch.exp(lv) + 1e-4)

    out = {}
    # dropout: inject fixed masks
    if dr_rate > 0:
        enc_w = list(H)
        dec_w = list(reversed(H))
        for i, w in enumerate(enc_w):
            m = (torch.rand(B, w) >= dr_rate)
            setattr(model.encoder.FC, f"D{i}", FixedDropout(dr_rate, m))
            out[f"mask/enc{i}"] = m.numpy().astype(np.uint8)
        m = (torch.rand(B, dec_w[0]) >= dr_rate)
        model.decoder.FirstL.D0 = FixedDropout(dr_rate, m)
        out["mask/dec0"] = m.numpy().astype(np.uint8)
        for i, w in enumerate(dec_w[1:], start=1):
            m = (torch.rand(B, w) >= dr_rate)
            setattr(model.decoder.HiddenL, f"D{i}", FixedDropout(dr_rate, m))
            out[f"mask/dec{i}"] = m.numpy().astype(np.uint8)

    for k, v in model.state_dict().items():
        out["w/" + k] = v.detach().numpy().copy()

    recon_l, kl, mmd_l = model(x=x, batch=batch, sizefactor=sizefactor)
    alpha = 0.37
    loss = recon_l + alpha * kl + mmd_l
    model.zero_grad()
    loss.backward()
    for k, p in model.named_parameters():
        out["g/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    out.update(recon=np.float32(recon_l.item()), kl=np.float32(kl.item()), mmd=np.float32(mmd_l.item()),
               alpha=np.float32(alpha))

    # BatchNorm running statistics after the single train-mode forward above
    for k, v in model.state_dict().items():
        if "running_" in k:
            out["bn_after/" + k] = v.detach().numpy().copy()

    # latent (eval mode, mean / mean_var) for get_latent parity
    model.eval()
    with torch.no_grad():
        zm, zv = model.get_latent(x, batch, mean=False, mean_var=True)
        y1 = model.get_y(x, batch)
    model.train()
    out.update(z_mean=zm.numpy(), z_var=zv.numpy(), y1=y1.numpy())

    # f64 re-evaluation of recon / kl (mmd alphas are float32-only in the reference)
    import copy
    m64 = copy.deepcopy(model).double()
    m64.sampling = lambda mu, lv: mu + eps.double() * torch.sqrt(torch.exp(lv) + 1e-4)
    m64.use_mmd = False
    torch.set_default_dtype(torch.float64)      # one_hot_encoder builds torch.zeros(...) in the default dtype
    try:
        r64, k64, _ = m64(x=x.double(), batch=batch, sizefactor=sizefactor.double())
    finally:
        torch.set_default_dtype(torch.float32)
    out.update(recon64=np.float64(r64.item()), kl64=np.float64(k64.item()))

    if adam_steps:
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=0.01,
                               weight_decay=wd)
        for _ in range(adam_steps):
            r, k, m = model(x=x, batch=batch, sizefactor=sizefactor)
            l = r + alpha * k + m
            opt.zero_grad()
            l.backward()
            opt.step()
        for k_, v in model.state_dict().items():
            out["adam/" + k_] = v.detach().numpy().copy()
        out["adam_steps"] = np.int32(adam_steps)
        out["adam_wd"] = np.float32(wd)
        out["adam_last_loss"] = np.float32(l.item())

    cfg = dict(G=G, C=C, H=list(H), Z=Z, B=B, recon=recon, use_bn=use_bn, use_ln=use_ln, use_mmd=use_mmd,
               mmd_on=mmd_on, boundary=boundary, beta=beta, dr_rate=dr_rate)
    out.update(cfg=json.dumps(cfg), x_data=xs.data.astype(np.float32), x_indices=xs.indices.astype(np.int32),
               x_indptr=xs.indptr.astype(np.int64), batch=cond, sizefactor=sizefactor.numpy(), eps=eps.numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: recon={recon_l.item():.6f} kl={kl.item():.6f} mmd={mmd_l.item():.6f} "
          f"(recon64={r64.item():.8f})")


def op_goldens(rng):
    """Op-level known-answer vectors straight from the reference loss functions."""
    out = {}
    # nb / zinb elementwise with gradients
    B, G = 12, 33
    x = torch.tensor(rng.poisson(rng.gamma(0.3, 3.0, size=(B, G))).astype(np.float32))
    x[0, :] = 0
    x[1, 3] = 2500.0
    mu = torch.tensor(rng.gamma(1.0, 2.0, size=(B, G)).astype(np.float32) + 1e-6, requires_grad=True)
    theta = torch.tensor(np.exp(rng.normal(0, 1.5, size=(B, G))).astype(np.float32), requires_grad=True)
    pi = torch.tensor(rng.normal(0, 2, size=(B, G)).astype(np.float32), requires_grad=True)
    ll = ref_losses.nb(x, mu, theta)
    ll.sum().backward()
    out.update(nb_x=x.numpy(), nb_mu=mu.detach().numpy(), nb_theta=theta.detach().numpy(), nb_ll=ll.detach().numpy(),
               nb_dmu=mu.grad.numpy().copy(), nb_dtheta=theta.grad.numpy().copy())
    mu.grad = None
    theta.grad = None
    zl = ref_losses.zinb(x, mu, theta, pi)
    zl.sum().backward()
    out.update(zinb_pi=pi.detach().numpy(), zinb_ll=zl.detach().numpy(), zinb_dmu=mu.grad.numpy().copy(),
               zinb_dtheta=theta.grad.numpy().copy(), zinb_dpi=pi.grad.numpy().copy())
    # f64 versions
    ll64 = ref_losses.nb(x.double(), mu.detach().double(), theta.detach().double())
    out["nb_ll64"] = ll64.numpy()

    # mmd: several label layouts
    cases = {
        "a": dict(B=40, D=10, C=4, boundary=None, beta=1.0),
        "b": dict(B=64, D=10, C=20, boundary=None, beta=0.5),    # some conditions empty / singletons
        "c": dict(B=48, D=16, C=5, boundary=3, beta=2.0),
        "d": dict(B=9, D=3, C=6, boundary=2, beta=1.0),          # singletons skipped by the >=2 rule
        "e": dict(B=130, D=10, C=3, boundary=None, beta=1.0),
    }
    for key, c in cases.items():
        y = torch.tensor(rng.normal(0, 1.2, size=(c["B"], c["D"])).astype(np.float32), requires_grad=True)
        lab = torch.tensor(rng.integers(0, c["C"], size=c["B"]).astype(np.int64))
        if key == "d":
            lab = torch.tensor([0, 0, 1, 2, 2, 2, 3, 5, 5])
        val = ref_losses.mmd(y, lab, c["C"], c["beta"], c["boundary"])
        if val.requires_grad:
            val.backward()
            g = y.grad.numpy().copy()
        else:
            g = np.zeros_like(y.detach().numpy())
        out[f"mmd_{key}_y"] = y.detach().numpy()
        out[f"mmd_{key}_c"] = lab.numpy()
        out[f"mmd_{key}_val"] = np.float32(val.item())
        out[f"mmd_{key}_grad"] = g
        out[f"mmd_{key}_cfg"] = json.dumps({k: v for k, v in c.items()})
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops: written")


def host_goldens(rng):
    """Host-logic fixtures: early stopping, split, dataset, stratifier weights, label encoding."""
    out = {}
    # early stopping state machine on a synthetic metric sequence
    seq = np.concatenate([np.linspace(10, 5, 25), 5 + 0.01 * rng.random(60)]).astype(np.float64)
    es = EarlyStopping()
    rec = []
    for v in seq:
        upd = es.update_state(v)
        cont, lr = quiet(es.step, v)
        rec.append((int(upd), int(cont), int(lr)))
        if not cont:
            break
    out["es_seq"] = seq
    out["es_rec"] = np.array(rec, dtype=np.int32)
    es2 = EarlyStopping(patience=5, lr_patience=2, threshold=0.05, lr_factor=0.5)
    seq2 = np.array([9, 8, 7, 6.99, 6.98, 6.97, 6.5, 6.49, 6.48, 6.47, 6.46, 6.45, 6.44, 6.43], dtype=np.float64)
    rec2 = []
    for v in seq2:
        upd = es2.update_state(v)
        cont, lr = quiet(es2.step, v)
        rec2.append((int(upd), int(cont), int(lr)))
        if not cont:
            break
    out["es2_seq"] = seq2
    out["es2_rec"] = np.array(rec2, dtype=np.int32)

    # split / dataset
    n, g, C = 57, 20, 4
    xs, cond = synth_counts(rng, n, g, C)
    names = np.array([f"study{i}" for i in range(C)])
    obs = pd.DataFrame({"study": names[cond]})
    ad = FakeAnnData(xs, obs)
    tr, va = train_test_split(ad, 0.9, condition_key="study")
    out["split_cond"] = cond
    out["split_train"] = tr.astype(np.int64)
    out["split_valid"] = va.astype(np.int64)
    enc = {nm: i for i, nm in enumerate(names)}
    ds = AnnotatedDataset(ad, condition_key="study", condition_encoder=enc)
    out["ds_x_data"] = xs.data.astype(np.float32)
    out["ds_x_indices"] = xs.indices.astype(np.int32)
    out["ds_x_indptr"] = xs.indptr.astype(np.int64)
    out["ds_shape"] = np.array(xs.shape, dtype=np.int64)
    out["ds_weights"] = ds.stratifier_weights
    out["ds_sizefactors"] = ds.size_factors.numpy()
    items = [ds[i] for i in (3, 10, 11, 40)]
    col = custom_collate(items)
    out["ds_collate_x"] = col["x"].numpy()
    out["ds_collate_sizefactor"] = col["sizefactor"].numpy()
    out["ds_collate_batch"] = col["batch"].numpy()
    out["ds_collate_labeled"] = col["labeled"].numpy()
    # label encoder with an unknown label
    enc_partial = {nm: i for i, nm in enumerate(names[:-1])}
    lab = quiet(label_encoder, ad, enc_partial, "study")
    out["le_partial"] = lab
    np.savez_compressed(os.path.join(HERE, "host.npz"), **out)
    print("host: written")


def surgery_golden(rng):
    """Reference TRVAE wrapper: build -> load_query_data (expand + freeze) -> query forward/backward."""
    G, C, H, Z = 48, 3, [32, 16], 5
    n_ref, n_q = 60, 40
    xs, cond = synth_counts(rng, n_ref, G, C)
    names = [f"ref{i}" for i in range(C)]
    ad = FakeAnnData(xs.toarray(), pd.DataFrame({"study": np.array(names)[cond]}))
    torch.manual_seed(11)
    ref = quiet(TRVAE, ad, condition_key="study", conditions=list(names), hidden_layer_sizes=H, latent_dim=Z,
                dr_rate=0.05, use_mmd=True, recon_loss="nb")
    out = {}
    for k, v in ref.model.state_dict().items():
        out["ref_w/" + k] = v.detach().numpy().copy()
    xq, cq = synth_counts(rng, n_q, G, 3)
    qnames = np.array(["ref1", "newA", "newB"])          # one known + two new conditions
    adq = FakeAnnData(xq.toarray(), pd.DataFrame({"study": qnames[cq]}))
    torch.manual_seed(12)
    q = quiet(TRVAE.load_query_data, adq, ref)
    out["q_conditions"] = np.array(q.conditions_).astype(str)
    out["q_trainable"] = np.array([n for n, p in q.model.named_parameters() if p.requires_grad]).astype(str)
    for k, v in q.model.state_dict().items():
        out["q_w/" + k] = v.detach().numpy().copy()
    out["q_dr_rate"] = np.float32(q.dr_rate_)
    # forward/backward on the query cells
    qb = torch.tensor(np.array([q.model.condition_encoder[s] for s in adq.obs["study"]]))
    x = torch.tensor(xq.toarray())
    sf = x.sum(1)
    eps = torch.randn(n_q, Z)
    q.model.sampling = lambda mu, lv: mu + eps * torch.sqrt(torch.exp(lv) + 1e-4)
    q.model.train()
    r, k, m = q.model(x=x, batch=qb, sizefactor=sf)
    (r + 0.5 * k + m).backward()
    for n_, p in q.model.named_parameters():
        if p.requires_grad:
            out["q_g/" + n_] = p.grad.numpy().copy()
    out.update(q_recon=np.float32(r.item()), q_kl=np.float32(k.item()), q_mmd=np.float32(m.item()),
               q_alpha=np.float32(0.5), q_x_data=xq.data.astype(np.float32), q_x_indices=xq.indices.astype(np.int32),
               q_x_indptr=xq.indptr.astype(np.int64), q_batch=qb.numpy(), q_eps=eps.numpy(),
               q_study=np.array(adq.obs["study"]).astype(str), ref_study=np.array(ad.obs["study"]).astype(str),
               cfg=json.dumps(dict(G=G, C=C, H=H, Z=Z)))
    # get_latent through the wrapper (mean=True) on the query cells
    lat = q.get_latent(adq.X, adq.obs["study"], mean=True)
    out["q_latent_mean"] = lat
    np.savez_compressed(os.path.join(HERE, "surgery.npz"), **out)
    print("surgery: trainable =", list(out["q_trainable"]))


def main():
    rng = np.random.default_rng(20240521)
    run_trvae_case("trvae_nb_ln_small", rng, G=64, C=3, H=[32, 16], Z=5, B=16, recon="nb", use_bn=False,
                   use_ln=True, use_mmd=True, mmd_on="z", boundary=None, beta=1.0, adam_steps=3, seed=1)
    run_trvae_case("trvae_nb_ln_g200_c20", rng, G=200, C=20, H=[128, 128], Z=10, B=128, recon="nb", use_bn=False,
                   use_ln=True, use_mmd=True, mmd_on="z", boundary=None, beta=1.0, seed=2)
    run_trvae_case("trvae_zinb_bn", rng, G=48, C=4, H=[24], Z=4, B=32, recon="zinb", use_bn=True,
                   use_ln=False, use_mmd=False, mmd_on="z", boundary=None, beta=1.0, seed=3)
    run_trvae_case("trvae_mse_plain_mmdy", rng, G=40, C=5, H=[32, 16], Z=6, B=40, recon="mse", use_bn=False,
                   use_ln=False, use_mmd=True, mmd_on="y", boundary=3, beta=2.0, seed=4)
    run_trvae_case("trvae_nb_dropout", rng, G=64, C=3, H=[32, 16], Z=5, B=16, recon="nb", use_bn=False,
                   use_ln=True, use_mmd=True, mmd_on="z", boundary=None, beta=1.0, dr_rate=0.2, seed=5)
    run_trvae_case("trvae_nb_ragged", rng, G=37, C=5, H=[20, 12, 8], Z=3, B=7, recon="nb", use_bn=False,
                   use_ln=True, use_mmd=True, mmd_on="z", boundary=None, beta=1.0,
                   cond_override=[0, 0, 0, 2, 4, 4, 0], zero_row=True, adam_steps=2, wd=0.0, seed=6)
    op_goldens(rng)
    host_goldens(rng)
    surgery_golden(rng)


if __name__ == "__main__":
    main()
