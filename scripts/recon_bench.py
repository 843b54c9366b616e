"""Dev probe: time the NB likelihood kernel alone (replayed from a recorded launch plan) over rows-per-workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_trvae_params, epoch_batches
from scarches_amd.engine import CVAEEngine
from scarches_amd.synth import make_synthetic_csr

dev = torch.device("cuda", 0)
G, C, Z, H = 2000, 20, 10, [128, 128]
indptr, indices, data, cond_all, _ = make_synthetic_csr(100_000, G, C, dev, seed=1000)
print("max count", data.max().item(), "frac > 16:", (data > 16).float().mean().item(), flush=True)
P = build_trvae_params(G, C, H, Z)
gen = torch.Generator(device=dev); gen.manual_seed(1)
for B in (8192,):
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    ids, cs = epoch_batches(cond_all, 2, B, gen)
    eng.train_step(ids[0], cs[0], B, 0.5, 1e-3, use_graph="plan")
    torch.cuda.synchronize()
    plan = next(iter(eng._plans.values()))["fwd"]
    for name in ("sca_recon_rows", "sca_csr_gather_rows"):
        hits = [(fn, args) for fn, args, _ in plan if getattr(fn, "__name__", "") == name]
        if not hits:
            print("no", name, [getattr(f, "__name__", "?") for f, _, _ in plan][:40]); continue
        fn, args = hits[0]
        variants = (4, 8) if name == "sca_recon_rows" else (None,)
        for rpb in variants:
            a = list(args)
            if rpb is not None:
                a[15] = rpb
            for _ in range(2):
                fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(*a)
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} {name} rpb={rpb}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
