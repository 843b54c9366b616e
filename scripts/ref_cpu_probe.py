"""Build-container-only probe: the REAL reference trVAE (imported from /root/reference through tests/golden/_ref_import)
timed on this container's CPU at the benchmark's cfg2 shape: forward + backward + Adam per minibatch (dense minibatch
already collated - the reference's per-cell scipy row slicing is NOT included).  The number is quoted as a constant in
bench.py's cpu_baseline object; the reference itself never travels to the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import _ref_import
_ref_import.install()
from scarches.models.trvae.trvae import trVAE
import contextlib, io

G, C, H, Z = 2000, 20, [128, 128], 10
for threads in (1, 4, 8):
    torch.set_num_threads(threads)
    for B in (128, 1024):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = trVAE(G, [f"c{i}" for i in range(C)], list(H), Z, 0.05, True, "z", None, "nb", 1.0, False, True)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=0.01, weight_decay=0.04)
        rng = np.random.default_rng(0)
        x = torch.tensor(rng.poisson(rng.gamma(0.3, 0.5, size=(B, G))).astype(np.float32))
        batch = torch.tensor(np.sort(rng.integers(0, C, size=B)))
        sf = x.sum(1)
        n, t0 = 0, None
        while True:
            r, kl, mmd = m(x=x, batch=batch, sizefactor=sf)
            loss = r + 0.5 * kl + mmd
            opt.zero_grad(); loss.backward(); opt.step()
            n += 1
            if n == 2:
                t0 = time.time(); n0 = n
            if t0 is not None and time.time() - t0 > 6:
                break
        dt = time.time() - t0
        print(f"threads={threads} B={B}: {(n - n0) * B / dt:.0f} cells/s ({dt / (n - n0) * 1e3:.1f} ms/step)", flush=True)
