"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRVAE_CASES = ["trvae_nb_ln_small", "trvae_nb_ln_g200_c20", "trvae_zinb_bn", "trvae_mse_plain_mmdy",
               "trvae_nb_dropout", "trvae_nb_ragged"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    W = {k[2:]: z[k] for k in z.files if k.startswith("w/")}
    Gr = {k[2:]: z[k] for k in z.files if k.startswith("g/")}
    masks = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("mask/")}
    return z, cfg, W, Gr, masks


def dense_from_case(z, G):
    import scipy.sparse as sp
    B = len(z["x_indptr"]) - 1
    m = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(B, G))
    return torch.tensor(m.toarray().astype(np.float32))


def rel_err(a, b):
    """max-norm relative error of tensor a against reference b."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


_REF64 = {}


def golden_ref64(name):
    """fp64 evaluation of the (pinned) oracle on a golden case's inputs: the yardstick against which the reference's
    own fp32 rounding noise is measured.  Returns (losses dict, gradients dict of numpy arrays)."""
    if name in _REF64:
        return _REF64[name]
    z, cfg, W, Gr, masks = load_case(name)
    x = dense_from_case(z, cfg["G"])
    sf, eps, alpha = torch.tensor(z["sizefactor"]), torch.tensor(z["eps"]), float(z["alpha"])
    if name.startswith("expimap"):
        from oracle import expimap_oracle as E
        mask = torch.tensor(z["mask"])
        ext_mask = torch.tensor(z["ext_mask"]) if "ext_mask" in z.files else None
        res, grads, _ = E.expimap_loss_and_grads({k: v for k, v in W.items() if not k.endswith(".mask")}, cfg, x,
                                                 z["batch"], sf, eps, alpha, float(z["beta"]), mask, ext_mask,
                                                 masks or None, dtype=torch.float64,
                                                 eps_l=torch.tensor(z["eps_l"]) if "eps_l" in z.files else None)
    elif name.startswith("scpoli"):
        from oracle import scpoli_oracle as S
        res, grads, _ = S.scpoli_loss_and_grads(W, cfg, x, z["batch"], z["combined"], sf, eps, alpha, float(z["eta"]),
                                                float(z["unlabeled_weight"]), z["celltypes"], z["labeled"],
                                                z["protos"], z["protos_u"], masks or None, dtype=torch.float64)
    else:
        from oracle import cvae_oracle as O
        res, grads, _ = O.trvae_loss_and_grads(W, cfg, x, z["batch"], sf, eps, alpha, masks or None,
                                               dtype=torch.float64)
    _REF64[name] = (res, {k: v.numpy() for k, v in grads.items()})
    return _REF64[name]


def grads_close(got: dict, ref: dict, rtol=1e-5, floor_frac=0.0, ref64: dict = None, k_noise=2.0, tag=None):
    """Per-tensor max-norm comparison at the north-star tolerance: |got - ref| <= rtol * scale + k_noise * noise, where
    `ref` is the reference's fp32 result, noise = max |ref - ref64| is the reference's own distance from the fp64
    yardstick (measured, 0 when no yardstick is given) and scale = max |ref|.  floor_frac (x the largest gradient
    entry of the whole model) covers mathematically-zero tensors (a bias in front of BatchNorm: both sides hold pure
    rounding residue of the order 1e-8 x the model's gradient scale, in an order that float atomics do not fix).
    With `tag` the achieved errors are printed next to the yardstick."""
    gmax = max(float(np.max(np.abs(v))) for v in ref.values())
    bad = []
    for k, r in ref.items():
        if k not in got:
            continue
        a = np.asarray(got[k], dtype=np.float64)
        r = np.asarray(r, dtype=np.float64)
        err = float(np.max(np.abs(a - r)))
        scale = float(np.max(np.abs(r)))
        noise = float(np.max(np.abs(r - np.asarray(ref64[k], dtype=np.float64)))) if ref64 is not None else 0.0
        bound = rtol * scale + k_noise * noise + floor_frac * gmax
        if tag:
            print("[%s] %-44s scale %.3e  err %.3e  ref32-ref64 %.3e  bound %.3e  (rel %.1e)%s"
                  % (tag, k, scale, err, noise, bound, err / max(scale, 1e-300), "" if err <= bound else "  FAIL"))
        if err > bound:
            bad.append((k, err, scale, noise))
    return bad
