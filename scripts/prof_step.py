import sys, time, torch
sys.path.insert(0, '.')
from scripts.dev_probe_lib import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng, cond, N = build(100000)
ids = torch.randint(0, N, (B,), device='cuda')
c = cond[ids]; order = torch.argsort(c, stable=True)
ids, c = ids[order].contiguous(), c[order].contiguous()
for it in range(3 + K):
    eng.begin_step(0.5, 1e-3); eng.forward_backward(ids, c, B); eng.adam_step()
torch.cuda.synchronize()
