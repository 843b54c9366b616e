"""Dev probe: ms/step of the cfg2 step for the launch modes (eager / plan / graph) at several batch sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_trvae_params, epoch_batches
from scarches_amd.engine import CVAEEngine
from scarches_amd.synth import make_synthetic_csr

dev = torch.device("cuda", 0)
G, C, Z, H = 2000, 20, 10, [128, 128]
indptr, indices, data, cond_all, _ = make_synthetic_csr(200_000, G, C, dev, seed=1000)
P = build_trvae_params(G, C, H, Z)
spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
            beta=1.0, dr_rate=0.05, seed=2020, fused_mid=os.environ.get("SCA_FUSED", "1") == "1")
gen = torch.Generator(device=dev); gen.manual_seed(1)
batches = [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else "128,1024,4096,16384".split(","))]
for B in batches:
    for mode, name, hp in ((False, "eager", False), ("plan", "plan", False), (True, "graph", False)):
        eng = CVAEEngine(spec, P, dev)
        eng.bind_csr(indptr, indices, data)
        steps, warm = 40, 8
        ids, cs = epoch_batches(cond_all, steps + warm, B, gen)
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        stream = torch.cuda.Stream(device=dev, priority=-1) if hp else torch.cuda.current_stream(dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for i in range(warm):
                eng.train_step(ids[i], cs[i], B, 0.5, 1e-3, use_graph=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(warm, warm + steps):
                eng.train_step(ids[i], cs[i], B, 0.5, 1e-3, use_graph=mode)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"B={B:6d} {name:6s} {dt * 1e3:7.3f} ms/step  {B / dt / 1e6:6.2f} Mcells/s  losses {eng.losses.tolist()}",
              flush=True)
