"""Dev tool: run-to-run spread of a short training trajectory (same seeds, same minibatches, fresh engines).
usage: python scripts/run_noise.py [B] [mode: plan|graph|eager] [runs] [steps]
Prints, per pair of runs, the largest parameter difference and the tensor it sits in."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import build_trvae_params  # noqa: E402
from scarches_amd.engine import CVAEEngine  # noqa: E402
from scarches_amd.synth import make_synthetic_csr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = {"plan": "plan", "graph": True, "eager": False}[sys.argv[2] if len(sys.argv) > 2 else "graph"]
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
G, C, Z, H, N = 2000, 20, 10, [128, 128], 60_000
dev = torch.device("cuda", 0)
indptr, indices, vals, cond_all, _ = make_synthetic_csr(N, G, C, dev, seed=91)
gen = torch.Generator(device="cuda")
gen.manual_seed(int(os.environ.get("SEED", "3")))
bs = []
for i in range(steps):
    rows = torch.randperm(N, generator=gen, device="cuda")[:B]
    cond = cond_all[rows]
    order = torch.argsort(cond, stable=True)
    bs.append((rows[order].contiguous(), cond[order].contiguous().to(torch.int32)))
spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
            beta=1.0, dr_rate=0.05, seed=2020)
outs = []
for r in range(runs):
    eng = CVAEEngine(spec, build_trvae_params(G, C, H, Z), "cuda:0")
    eng.bind_csr(indptr, indices, vals)
    traj = []
    for rows, cond in bs:
        eng.train_step(rows, cond, B, 0.5, 1e-3, use_graph=mode)
        traj.append(eng.flat_p.detach().clone())
    torch.cuda.synchronize()
    outs.append((traj, eng))
eng = outs[0][1]
for r in range(1, runs):
    line = []
    for s in range(steps):
        d = (outs[r][0][s] - outs[0][0][s]).abs()
        i = int(d.argmax())
        name = next((k for k in eng.names if eng.offsets[k] <= i < eng.offsets[k] + int(np.prod(eng.shapes[k]))), "?")
        line.append("%.1e@%s" % (float(d.max()), name))
    print("run", r, "vs 0:", " ".join(line))
    d = (outs[r][0][-1] - outs[0][0][-1]).abs()
    if float(d.max()) > 1e-6:
        print("    tensors off by > 1e-6:", [(k, "%.1e" % float(d[eng.offsets[k]:eng.offsets[k] + int(np.prod(eng.shapes[k]))].max()),
                                             int((d[eng.offsets[k]:eng.offsets[k] + int(np.prod(eng.shapes[k]))] > 1e-6).sum()))
                                            for k in eng.names
                                            if float(d[eng.offsets[k]:eng.offsets[k] + int(np.prod(eng.shapes[k]))].max()) > 1e-6])
print("max |p| =", float(outs[0][0][-1].abs().max()))
