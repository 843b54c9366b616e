import sys, time, torch
sys.path.insert(0, '.')
from collections import OrderedDict
from scarches_amd.synth import make_synthetic_csr
from scarches_amd.engine import CVAEEngine
dev = 'cuda'
N, G, C = 200000, 2000, 20
t0 = time.time()
indptr, indices, data, cond, ct = make_synthetic_csr(N, G, C, dev, seed=0)
torch.cuda.synchronize()
print('gen', time.time() - t0, 'nnz/cell', data.numel() / N, 'density', data.numel() / N / G, 'max', data.max().item())
H, Z = [128, 128], 10
def mk(shape, s=0.05): return torch.randn(*shape) * s
P = OrderedDict()
P['theta'] = torch.randn(G, C)
P['encoder.FC.L0.expr_L.weight'] = mk((128, G), 0.02); P['encoder.FC.L0.expr_L.bias'] = mk((128,))
P['encoder.FC.L0.cond_L.weight'] = mk((128, C))
P['encoder.FC.L1.weight'] = mk((128, 128)); P['encoder.FC.L1.bias'] = mk((128,))
P['encoder.mean_encoder.weight'] = mk((Z, 128)); P['encoder.mean_encoder.bias'] = mk((Z,))
P['encoder.log_var_encoder.weight'] = mk((Z, 128)); P['encoder.log_var_encoder.bias'] = mk((Z,))
P['decoder.FirstL.L0.expr_L.weight'] = mk((128, Z)); P['decoder.FirstL.L0.cond_L.weight'] = mk((128, C))
P['decoder.HiddenL.L1.weight'] = mk((128, 128))
P['decoder.mean_decoder.0.weight'] = mk((G, 128)); P['decoder.mean_decoder.0.bias'] = mk((G,))
spec = dict(G=G, C=C, H=H, Z=Z, recon='nb', use_ln=True, use_mmd=True, mmd_on='z', boundary=None, beta=1.0, dr_rate=0.05)
eng = CVAEEngine(spec, P, dev)
eng.bind_csr(indptr, indices, data)
for B in (128, 1024, 4096, 16384):
    ids = torch.randint(0, N, (B,), device=dev)
    c = cond[ids]
    order = torch.argsort(c, stable=True)
    ids, c = ids[order].contiguous(), c[order].contiguous()
    for it in range(3):
        eng.begin_step(0.5, 1e-3); eng.forward_backward(ids, c, B); eng.adam_step()
    torch.cuda.synchronize()
    K = 20
    t0 = time.time()
    for it in range(K):
        eng.begin_step(0.5, 1e-3); eng.forward_backward(ids, c, B); eng.adam_step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    print(f'B={B} eager {dt*1e3:.3f} ms/step  {B/dt/1e6:.3f} Mcells/s  losses', eng.losses.tolist())
    for it in range(3):
        eng.train_step(ids, c, B, 0.5, 1e-3)
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(K):
        eng.train_step(ids, c, B, 0.5, 1e-3)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    print(f'B={B} graph {dt*1e3:.3f} ms/step  {B/dt/1e6:.3f} Mcells/s  losses', eng.losses.tolist())
