"""Dev tool: repeat ONE forward/backward from identical parameters / RNG counters and report which workspace buffers
and gradients differ between repeats.  usage: python scripts/run_noise2.py [B] [repeats] [batch index]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import build_trvae_params  # noqa: E402
from scarches_amd.engine import CVAEEngine  # noqa: E402
from scarches_amd.synth import make_synthetic_csr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bi = int(sys.argv[3]) if len(sys.argv) > 3 else 5
G, C, Z, H, N = 2000, 20, 10, [128, 128], 60_000
dev = torch.device("cuda", 0)
indptr, indices, vals, cond_all, _ = make_synthetic_csr(N, G, C, dev, seed=91)
gen = torch.Generator(device="cuda")
gen.manual_seed(3)
bs = []
for i in range(bi + 1):
    rows = torch.randperm(N, generator=gen, device="cuda")[:B]
    cond = cond_all[rows]
    order = torch.argsort(cond, stable=True)
    bs.append((rows[order].contiguous(), cond[order].contiguous().to(torch.int32)))
spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
            beta=1.0, dr_rate=0.05, seed=2020)
eng = CVAEEngine(spec, build_trvae_params(G, C, H, Z), "cuda:0")
eng.bind_csr(indptr, indices, vals)
for rows, cond in bs[:bi]:
    eng.train_step(rows, cond, B, 0.5, 1e-3, use_graph=False)
torch.cuda.synchronize()
p0, ctr0, t0 = eng.flat_p.clone(), eng.step_ctr.clone(), eng.adam_t
rows, cond = bs[bi]
keys = None
TOL = float(os.environ.get('TOL', '1e-5'))
ref = None
for r in range(reps):
    eng.flat_p.copy_(p0); eng.step_ctr.copy_(ctr0); eng.adam_t = t0
    if r > 0 and os.environ.get("NOISE"):
        eng.flat_p.mul_(1.0 + float(os.environ["NOISE"]) * torch.randn_like(eng.flat_p))
    eng.begin_step(0.5, 1e-3)
    eng.forward_backward(rows, cond, B)
    torch.cuda.synchronize()
    snap = {k: v.detach().clone() for k, v in eng.ws.items() if torch.is_tensor(v) and v.dtype == torch.float32}
    snap["flat_g"] = eng.flat_g.clone()
    snap["losses"] = eng.losses.clone()
    if ref is None:
        ref = snap
        continue
    bad = []
    for k, v in snap.items():
        a, b = v.flatten(), ref[k].flatten()
        n = min(a.numel(), b.numel())
        d = (a[:n] - b[:n]).abs()
        d = torch.nan_to_num(d, nan=0.0)
        m = float(d.max()) if n else 0.0
        sc = float(b[:n].abs().max()) if n else 0.0
        if m > TOL * max(sc, 1e-30):
            bad.append("%s %.1e/%.1e n=%d first=%d" % (k, m, sc, int((d > TOL * sc).sum()), int((d > TOL * sc).nonzero()[0])))
    print("rep", r, "differs:" if bad else "same", "; ".join(bad)[:1500])
