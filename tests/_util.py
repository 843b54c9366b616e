"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRVAE_CASES = ["trvae_nb_ln_small", "trvae_nb_ln_cfg2", "trvae_zinb_bn", "trvae_mse_plain_mmdy",
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


def grads_close(got: dict, ref: dict, rtol=5e-5, floor_frac=1e-6):
    """Per-tensor max-norm comparison; tensors that are mathematically zero (e.g. a bias in front of
    BatchNorm) are judged against floor_frac x the largest gradient entry of the whole model."""
    gmax = max(float(np.max(np.abs(v))) for v in ref.values())
    bad = []
    for k, r in ref.items():
        if k not in got:
            continue
        a = np.asarray(got[k], dtype=np.float64)
        r = np.asarray(r, dtype=np.float64)
        err = float(np.max(np.abs(a - r)))
        if err > rtol * float(np.max(np.abs(r))) + floor_frac * gmax:
            bad.append((k, err, float(np.max(np.abs(r)))))
    return bad
