"""Dev probe: fused-middle path vs one-launch-per-layer path on one odd configuration - where do the saved activations
differ, and are the differences ReLU gates of pre-activations that sit within rounding of zero?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import build_trvae_params
from scarches_amd.engine import CVAEEngine
from scarches_amd.synth import make_synthetic_csr
Gx, Cx, Hx, Zx, Bx, dr = 1001, 4, [128, 128, 128, 128], 16, 4097, 0.2
dev = torch.device("cuda", 0)
indptr, indices, vals, cond_all, _ = make_synthetic_csr(6000, Gx, Cx, dev, seed=3)
rows = torch.arange(Bx, device=dev) % 6000
order = torch.argsort(cond_all[rows], stable=True)
rows = rows[order].contiguous()
cond = cond_all[rows].to(torch.int32).contiguous()
eps = torch.randn(Bx, Zx, device=dev)
engs = []
for fused in (True, False):
    spec = dict(G=Gx, C=Cx, H=Hx, Z=Zx, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=dr, seed=11, fused_mid=fused)
    e = CVAEEngine(spec, build_trvae_params(Gx, Cx, Hx, Zx), dev)
    e.bind_csr(indptr, indices, vals)
    e.begin_step(alpha_kl=0.3, lr=1e-3)
    e.forward_backward(rows, cond, Bx, eps=eps)
    engs.append(e)
torch.cuda.synchronize()
a, b = engs
for k in sorted(a.ws):
    if k in b.ws and a.ws[k].shape == b.ws[k].shape and a.ws[k].dtype == torch.float32 and (".yn" in k or ".h" in k or k in ("z", "mu_lv")):
        x, y = a.ws[k][:Bx], b.ws[k][:Bx]
        d = (x - y).abs()
        flips = ((x > 0) != (y > 0)).sum().item() if ".yn" in k else -1
        print(f"{k:12s} max|diff| {d.max().item():.3e}  scale {y.abs().max().item():.3e}  gate flips {flips}"
              + (f"  |yn| at the flips {y[(x > 0) != (y > 0)].abs().tolist()[:5]}" if flips > 0 else ""))
for k in a.names:
    d = (a.g[k] - b.g[k]).abs().max().item()
    print(f"grad {k:40s} {d:.3e} / {b.g[k].abs().max().item():.3e}")
