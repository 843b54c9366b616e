import sys, time, os, torch
sys.path.insert(0, '.')
from scripts.dev_probe_lib import build
eng, cond, N = build(200000)
for B in (4096, 16384):
    ids = torch.randint(0, N, (B,), device='cuda'); c = cond[ids]; o = torch.argsort(c, stable=True); ids, c = ids[o].contiguous(), c[o].contiguous()
    for it in range(3): eng.train_step(ids, c, B, 0.5, 1e-3, use_graph=False)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(20): eng.train_step(ids, c, B, 0.5, 1e-3, use_graph=False)
    torch.cuda.synchronize(); print(os.environ.get('SCA_TAG', ''), f'B={B} eager {(time.time()-t0)/20*1e3:.3f} ms/step')
