"""Parity of the HIP step against the (pinned) CPU oracle, LIVE, at the shapes of BASELINE.json's configs - the widths
the golden fixtures (B <= 128, G <= 256) do not reach:

  cfg1  trVAE   G = 1000, C = 5,   H = [128, 128], Z = 10          (pancreas-shaped reference build)
  cfg2  trVAE   G = 2000, C = 20,  H = [128, 128], Z = 10, B in {128, 1024}, dropout masks injected (headline config)
  cfg5  trVAE   G = 2000, C = 120, H = [128, 128], Z = 10          (HLCA-shaped: > 100 studies)
  cfg3  expiMap G = 5000, L = 280 gene programs (mask density ~1.2 %), H = [256, 256, 256], C = 4
  cfg4  scPoli  G = 4000, 200 combined conditions (two covariates), H = [64], E = 10 + 5, Z = 10, 30 cell types

Tolerance (north_star: per-batch losses and gradients within rel 1e-5 (fp32) of the reference CPU path).  The oracle is
evaluated twice on the same inputs, in fp32 (= the reference path) and in fp64 (the yardstick).  For every loss and every
parameter-gradient tensor the test computes
    err   = max |hip - ref32|          noise = max |ref32 - ref64|  (the reference's own fp32 rounding, measured here)
    scale = max |ref64|
and asserts  err <= 1e-5 * scale + 2 * noise :  rel 1e-5 wherever the reference itself is that accurate, and never more
than twice the reference's own measured distance from the exact value where it is not (two fp32 evaluations with
different summation orders are each `noise` away from the exact value).  The achieved numbers are printed per tensor
(`pytest -s`, or the captured output of a failure) next to the yardstick.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

RTOL = 1e-5
K_NOISE = 2.0


# --------------------------------------------------------------------------------------------- comparison helpers
def compare(tag, hip, ref32, ref64, names=None):
    """Rows (name, scale, err, err64, noise, bound, ok) for dicts of arrays / scalars."""
    rows = []
    for k in (names or ref64.keys()):
        a = np.asarray(hip[k], dtype=np.float64)
        r32 = np.asarray(ref32[k], dtype=np.float64)
        r64 = np.asarray(ref64[k], dtype=np.float64)
        scale = float(np.max(np.abs(r64))) if r64.size else 0.0
        err = float(np.max(np.abs(a - r32))) if r64.size else 0.0
        err64 = float(np.max(np.abs(a - r64))) if r64.size else 0.0
        noise = float(np.max(np.abs(r32 - r64))) if r64.size else 0.0
        bound = RTOL * scale + K_NOISE * noise
        rows.append((k, scale, err, err64, noise, bound, err <= bound))
    print("\n[%s] %-44s %10s %10s %10s %10s %10s" % (tag, "tensor", "scale", "hip-ref32", "hip-ref64", "ref32-ref64",
                                                    "bound"))
    for k, scale, err, err64, noise, bound, ok in rows:
        print("[%s] %-44s %10.3e %10.3e %10.3e %10.3e %10.3e %s  (rel %.1e)"
              % (tag, k, scale, err, err64, noise, bound, "ok" if ok else "FAIL", err / max(scale, 1e-300)))
    return rows


def assert_rows(rows):
    bad = [(k, "err %.3e > bound %.3e (scale %.3e, noise %.3e)" % (err, bound, scale, noise))
           for k, scale, err, err64, noise, bound, ok in rows if not ok]
    assert not bad, bad


def synth_batch(G, C, B, seed, n_cells=6000):
    """A condition-sorted minibatch of synthetic NB counts: device CSR + (rows, cond) and the dense CPU copy."""
    from scarches_amd.synth import make_synthetic_csr
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, ctype = make_synthetic_csr(n_cells, G, C, dev, seed=seed)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1)
    rows = torch.randperm(n_cells, generator=gen, device=dev)[:B]
    cond = cond_all[rows]
    order = torch.argsort(cond, stable=True)
    rows, cond = rows[order].contiguous(), cond[order].contiguous().to(torch.int32)
    dense = torch.sparse_csr_tensor(indptr, indices.long(), vals, size=(n_cells, G)).to_dense()[rows].cpu()
    return dict(csr=(indptr, indices, vals), rows=rows, cond=cond, x=dense, sf=dense.sum(1),
                ctype=ctype[rows].contiguous())


def dropout_masks(tags_widths, B, p, seed):
    g = torch.Generator().manual_seed(seed)
    return {t: (torch.rand(B, w, generator=g) > p) for t, w in tags_widths}


def lin(a, b, bias=True):
    l = nn.Linear(a, b, bias=bias)
    return l.weight.data, (l.bias.data if bias else None)


# --------------------------------------------------------------------------------------------- trVAE (cfg1, cfg2, cfg5)
TRVAE_SHAPES = [
    pytest.param(dict(G=1000, C=5, H=[128, 128], Z=10, B=128, dr=0.05), id="cfg1-G1000-C5-B128"),
    pytest.param(dict(G=2000, C=20, H=[128, 128], Z=10, B=128, dr=0.05), id="cfg2-G2000-C20-B128"),
    pytest.param(dict(G=2000, C=20, H=[128, 128], Z=10, B=1024, dr=0.05), id="cfg2-G2000-C20-B1024"),
    pytest.param(dict(G=2000, C=20, H=[256, 64], Z=10, B=512, dr=0.05), id="cfg2-classdefault-256-64-B512"),
    pytest.param(dict(G=2000, C=120, H=[128, 128], Z=10, B=1024, dr=0.0), id="cfg5-G2000-C120-B1024"),
]


@pytest.mark.parametrize("shape", TRVAE_SHAPES)
@pytest.mark.parametrize("fused", [True, False], ids=["fusedmid", "perlayer"])
def test_trvae_step_at_config_shapes(shape, fused):
    from bench import build_trvae_params
    from oracle import cvae_oracle as O
    from scarches_amd.engine import CVAEEngine
    G, C, H, Z, B, dr = (shape[k] for k in ("G", "C", "H", "Z", "B", "dr"))
    d = synth_batch(G, C, B, seed=G + C + B)
    P = build_trvae_params(G, C, H, Z, seed=7)
    cfg = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
               beta=1.0, dr_rate=dr)
    eng = CVAEEngine(dict(cfg, seed=2020, fused_mid=fused), P, "cuda:0")
    assert eng.use_fused_mid == fused
    eng.bind_csr(*d["csr"])
    eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(3))
    Hd = list(reversed(H))
    masks = dropout_masks([(f"enc{i}", h) for i, h in enumerate(H)] + [(f"dec{i}", h) for i, h in enumerate(Hd)],
                          B, dr, seed=5) if dr > 0 else {}
    alpha = 0.35
    eng.set_hyper(alpha_kl=alpha)
    eng.forward_backward(d["rows"], d["cond"], B, eps=eps.cuda(),
                         masks={k: v.to(torch.uint8).cuda().contiguous() for k, v in masks.items()})
    torch.cuda.synchronize()
    Lh = eng.losses.cpu().numpy()
    Wnp = {k: v.numpy() for k, v in P.items()}
    out = {}
    for dt in (torch.float32, torch.float64):
        res, grads, _ = O.trvae_loss_and_grads(Wnp, cfg, d["x"], d["cond"].cpu().numpy(), d["sf"], eps, alpha,
                                               masks or None, dtype=dt)
        out[dt] = (res, {k: v.numpy() for k, v in grads.items()})
    tag = "trvae G%d C%d B%d %s" % (G, C, B, "fused" if fused else "perlayer")
    hipL = dict(recon=Lh[0], kl=Lh[1], mmd=Lh[2], loss=Lh[3])
    rows = compare(tag, hipL, out[torch.float32][0], out[torch.float64][0], ["recon", "kl", "mmd", "loss"])
    rows += compare(tag, {k: eng.g[k].cpu().numpy() for k in eng.names}, out[torch.float32][1],
                    out[torch.float64][1])
    assert_rows(rows)


# --------------------------------------------------------------------------------------------- expiMap (cfg3)
def gene_program_mask(G, L, seed):
    """Binary [L, G] mask: program sizes LogNormal(median 30) clipped to [13, 300], every gene in >= 1 program
    (SURVEY 8(d) cfg3)."""
    rng = np.random.RandomState(seed)
    M = np.zeros((L, G), dtype=np.float32)
    sizes = np.clip(np.exp(rng.normal(np.log(30.0), 0.8, L)).astype(int), 13, 300)
    for l in range(L):
        M[l, rng.choice(G, sizes[l], replace=False)] = 1
    for g in np.where(M.sum(0) == 0)[0]:
        M[rng.randint(L), g] = 1
    return M


@pytest.mark.parametrize("shape", [
    pytest.param(dict(G=5000, L=280, H=[256, 256, 256], C=4, B=256, soft=False), id="cfg3-G5000-L280-hard-B256"),
    pytest.param(dict(G=5000, L=280, H=[256, 256, 256], C=4, B=512, soft=True), id="cfg3-G5000-L280-soft-B512"),
    pytest.param(dict(G=5000, L=280, H=[256, 256, 256], C=4, B=1000, soft=False), id="cfg3-G5000-L280-hard-B1000"),
])
def test_expimap_step_at_config_shape(shape):
    from oracle import expimap_oracle as XO
    from scarches_amd.engine import CVAEEngine
    G, L, H, C, B, soft = (shape[k] for k in ("G", "L", "H", "C", "B", "soft"))
    d = synth_batch(G, C, B, seed=31)
    M = gene_program_mask(G, L, seed=1)
    assert 0.005 < M.mean() < 0.03
    torch.manual_seed(11)
    P = OrderedDict()
    P["theta"] = torch.randn(G, C)
    sizes = [G] + H
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        if i == 0:
            P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"] = lin(a, b)
            P["encoder.FC.L0.cond_L.weight"] = lin(C, b, bias=False)[0]
        else:
            P[f"encoder.FC.L{i}.weight"], P[f"encoder.FC.L{i}.bias"] = lin(a, b)
    P["encoder.mean_encoder.weight"], P["encoder.mean_encoder.bias"] = lin(H[-1], L)
    P["encoder.log_var_encoder.weight"], P["encoder.log_var_encoder.bias"] = lin(H[-1], L)
    W = lin(L, G, bias=False)[0]
    P["decoder.L0.expr_L.weight"] = W if soft else W * torch.tensor(M.T)       # hard mask: zeroed at init
    P["decoder.L0.cond_L.weight"] = lin(C, G, bias=False)[0]
    cfg = dict(G=G, C=C, H=H, L=L, recon="nb", use_ln=True, dr_rate=0.0, n_ext=0, n_ext_m=0, soft_mask=soft,
               soft_ext_mask=False, use_hsic=False, hsic_one_vs_all=False)
    spec = dict(arch="expimap", G=G, C=C, H=H, Z=L, recon="nb", use_bn=False, use_ln=True, dr_rate=0.0, n_ext=0,
                n_ext_m=0, mask=None if soft else torch.tensor(M.T.copy()), ext_mask=None, use_hsic=False)
    eng = CVAEEngine(spec, P, "cuda:0")
    # Reactome-like hard masks go through the tile-skipping decoder kernels, soft masks through the dense GEMMs
    assert (eng.masked is None) == soft
    if not soft:
        assert eng.masked["skipped"] > 0.5, eng.masked["skipped"]
    eng.bind_csr(*d["csr"])
    eps = torch.randn(B, L, generator=torch.Generator().manual_seed(3))
    alpha = 0.5
    eng.set_hyper(alpha_kl=alpha)
    eng.forward_backward(d["rows"], d["cond"], B, eps=eps.cuda())
    torch.cuda.synchronize()
    Lh = eng.losses.cpu().numpy()
    Wnp = {k: v.numpy() for k, v in P.items()}
    out = {}
    for dt in (torch.float32, torch.float64):
        res, grads, _ = XO.expimap_loss_and_grads(Wnp, cfg, d["x"], d["cond"].cpu().numpy(), d["sf"], eps, alpha,
                                                  mask=None if soft else M, dtype=dt)
        out[dt] = (res, {k: v.numpy() for k, v in grads.items()})
    tag = "expimap G%d L%d B%d %s" % (G, L, B, "soft" if soft else "hard")
    rows = compare(tag, dict(recon=Lh[0], kl=Lh[1]), out[torch.float32][0], out[torch.float64][0], ["recon", "kl"])
    got = {k: eng.g[k].cpu().numpy() for k in eng.names}
    rows += compare(tag, got, out[torch.float32][1], out[torch.float64][1])
    assert_rows(rows)
    if not soft:
        assert np.all(got["decoder.L0.expr_L.weight"][M.T == 0] == 0)


# --------------------------------------------------------------------------------------------- scPoli (cfg4)
@pytest.mark.parametrize("B", [128, 1024])
def test_scpoli_step_at_config_shape(B):
    from oracle import scpoli_oracle as SO
    from scarches_amd.engine import CVAEEngine
    G, H, Z, E, n_conds, K = 4000, [64], 10, [10, 5], [40, 5], 30
    Cc = n_conds[0] * n_conds[1]                       # 200 combined conditions
    d = synth_batch(G, Cc, B, seed=41, n_cells=8000)
    combined = d["cond"]
    covs = torch.stack([combined // n_conds[1], combined % n_conds[1]], 1).to(torch.int32).contiguous()
    torch.manual_seed(13)
    P = OrderedDict()
    for i, (n, e) in enumerate(zip(n_conds, E)):
        P[f"embeddings.{i}.weight"] = torch.randn(n, e)
    P["theta"] = torch.randn(G, Cc)
    Et = sum(E)
    P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"] = lin(G, H[0])
    P["encoder.FC.L0.cond_L.weight"] = lin(Et, H[0], bias=False)[0]
    P["encoder.mean_encoder.weight"], P["encoder.mean_encoder.bias"] = lin(H[-1], Z)
    P["encoder.log_var_encoder.weight"], P["encoder.log_var_encoder.bias"] = lin(H[-1], Z)
    P["decoder.FirstL.L0.expr_L.weight"] = lin(Z, H[0], bias=False)[0]
    P["decoder.FirstL.L0.cond_L.weight"] = lin(Et, H[0], bias=False)[0]
    P["decoder.mean_decoder.0.weight"], P["decoder.mean_decoder.0.bias"] = lin(H[0], G)
    cfg = dict(G=G, H=H, Z=Z, E=E, n_conds=n_conds, recon="nb", use_ln=True, dr_rate=0.0, inject=["encoder", "decoder"],
               max_norm=1.0, C_combined=Cc)
    spec = dict(arch="scpoli", G=G, C=Cc, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, dr_rate=0.0, E=E,
                n_conds=n_conds, inject=cfg["inject"], max_norm=1.0)
    Wnp = {k: v.numpy().copy() for k, v in P.items()}         # the lookup renormalises embedding rows in place
    eng = CVAEEngine(spec, P, "cuda:0")
    eng.bind_csr(*d["csr"])
    eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(3))
    g = torch.Generator().manual_seed(17)
    protos = torch.randn(K, Z, generator=g)
    celltypes = (d["ctype"].cpu().long() % K).view(B, 1)
    celltypes[torch.rand(B, generator=g) < 0.2] = -1            # unknown labels
    labeled = (torch.rand(B, generator=g) < 0.8)
    alpha, eta = 0.5, 5.0
    eng.set_hyper(alpha_kl=alpha)
    proto = dict(P=protos.cuda(), Pu=None, celltypes=celltypes.to(torch.int32).cuda().contiguous(),
                 labeled=labeled.to(torch.uint8).cuda(), eta=eta, unlabeled_weight=0.0)
    eng.forward_backward(d["rows"], combined, B, eps=eps.cuda(), covs=covs, proto=proto)
    torch.cuda.synchronize()
    Lh = eng.losses.cpu().numpy()
    out = {}
    for dt in (torch.float32, torch.float64):
        res, grads, _ = SO.scpoli_loss_and_grads({k: v.copy() for k, v in Wnp.items()}, cfg, d["x"],
                                                 covs.cpu().numpy(), combined.cpu().numpy(), d["sf"], eps, alpha, eta,
                                                 0.0, celltypes.numpy(), labeled.numpy().astype(np.int64),
                                                 protos.numpy(), None, dtype=dt)
        res = dict(res, proto=eta * res["labeled"])
        out[dt] = (res, {k: v.numpy() for k, v in grads.items()})
    tag = "scpoli G%d C%d B%d" % (G, Cc, B)
    rows = compare(tag, dict(recon=Lh[0], kl=Lh[1], proto=Lh[2], loss=Lh[3]), out[torch.float32][0],
                   out[torch.float64][0], ["recon", "kl", "proto", "loss"])
    rows += compare(tag, {k: eng.g[k].cpu().numpy() for k in eng.names}, out[torch.float32][1],
                    out[torch.float64][1])
    assert_rows(rows)
