"""Throughput of the other two model families through the public API on BASELINE.json-shaped synthetic data
(configs[2]: expiMap, 5000 genes x 280 gene programs, hidden [256, 256, 256]; configs[3]: scPoli, 4000 genes, hidden
[64], embedding dim 10, 200 samples as conditions, 30 cell types).  Train cells / s = cells of the extra epochs /
extra wall time between a short and a longer `train()` call of the same model class (set-up, upload and plan recording
cancel).  Numbers go to DESIGN.md section 8; bench.py stays the trVAE north-star line.

    python scripts/bench_families.py [expimap|scpoli] [n_cells]
"""
import os
import sys
import time

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scarches_amd as sca                                   # noqa: E402
from scarches_amd.synth import make_synthetic_csr            # noqa: E402


def synth(n, G, C, T, seed):
    indptr, indices, data, cond, ctype = make_synthetic_csr(n, G, C, "cuda:0", seed=seed, n_types=T)
    X = sp.csr_matrix((data.cpu().numpy(), indices.cpu().numpy(), indptr.cpu().numpy()), shape=(n, G))
    obs = pd.DataFrame({"study": np.char.add("s", cond.cpu().numpy().astype(str)),
                        "cell_type": np.char.add("t", ctype.cpu().numpy().astype(str))})
    return sca.CellData(X, obs)


def timed(make, train, e_short=2, e_long=32):
    out = []
    for e in (e_short, e_long):
        m = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train(m, e)
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return (out[1] - out[0]) / (e_long - e_short)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "expimap"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    if which == "expimap":
        G, L = 5000, 280
        ad = synth(n, G, 4, 12, 1)
        rng = np.random.default_rng(0)
        sizes = np.clip(rng.lognormal(np.log(30), 0.8, L), 13, 300).astype(int)
        mask = np.zeros((G, L), dtype=np.float32)
        for k, s in enumerate(sizes):
            mask[rng.choice(G, s, replace=False), k] = 1
        mask[np.arange(G), rng.integers(0, L, G)] = np.maximum(mask[np.arange(G), rng.integers(0, L, G)], 1)
        ad.varm["I"] = mask
        make = lambda: sca.models.EXPIMAP(ad, condition_key="study", hidden_layer_sizes=[256, 256, 256],
                                          recon_loss="nb")
        train = lambda m, e: m.train(n_epochs=e, alpha=0.7, alpha_kl=0.5, weight_decay=0., batch_size=B, seed=1,
                                     use_early_stopping=False, monitor=False)
        desc = f"expiMap {n} x {G}, {L} gene programs (mask density {mask.mean():.3f}), hidden [256,256,256], 4 conditions"
    else:
        G = 4000
        ad = synth(n, G, 200, 30, 2)
        np.random.seed(0)
        make = lambda: sca.models.scPoli(ad, condition_keys="study", cell_type_keys="cell_type",
                                         hidden_layer_sizes=[64], latent_dim=10, embedding_dims=10, recon_loss="nb")
        train = lambda m, e: m.train(n_epochs=e, pretraining_epochs=max(1, e - 1), eta=5, batch_size=B, seed=1,
                                     use_early_stopping=False, monitor=False, unlabeled_prototype_training=False)
        desc = f"scPoli {n} x {G}, hidden [64], E=10, Z=10, 200 conditions, 30 cell types (last epoch with prototypes)"
    dt = timed(make, train)
    n_train = int(0.9 * n)
    if which == "expimap":
        mk = make().model.engine().masked
        desc += (", decoder: dense GEMMs" if mk is None else
                 f", decoder: tile-skipping (Kmax {mk['Kmax']} of {L} programs per 32-gene tile, "
                 f"{100 * mk['skipped']:.1f} % of the k-steps skipped)")
    print(f"{desc}: batch {B}: {dt:.3f} s / epoch (train {n_train} + validation {n - n_train} cells) = "
          f"{n / dt / 1e6:.2f} M cells/s through the public API", flush=True)


if __name__ == "__main__":
    main()
