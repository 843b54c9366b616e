"""Dev probe: get_latent throughput of the engine (eval-mode encoder pass: CSR gather -> encoder -> heads -> mean),
cfg2 shapes, cells resident in HBM."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trvae_params
from scarches_amd.engine import CVAEEngine
from scarches_amd.synth import make_synthetic_csr
dev = torch.device("cuda", 0)
G, C, Z, H, N = 2000, 20, 10, [128, 128], 500_000
indptr, indices, data, cond_all, _ = make_synthetic_csr(N, G, C, dev, seed=1000)
spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
            beta=1.0, dr_rate=0.05, seed=2020)
eng = CVAEEngine(spec, build_trvae_params(G, C, H, Z), dev)
eng.bind_csr(indptr, indices, data)
rows = torch.arange(N, device=dev)
out = torch.empty(N, Z, device=dev)
for B in (512, 8192, 65536):
    def sweep():
        for lo in range(0, N, B):
            n = min(B, N - lo)
            r = eng.encode(rows[lo:lo + n], cond_all[lo:lo + n], n, mean=True)
            out[lo:lo + n].copy_(r[0][:n])
    sweep(); torch.cuda.synchronize()
    t0 = time.perf_counter(); sweep(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"get_latent chunk {B}: {N / dt / 1e6:.2f} M cells/s ({dt * 1e3:.1f} ms for {N} cells)", flush=True)
