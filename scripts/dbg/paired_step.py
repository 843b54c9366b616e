"""Debug: first steps of the paired kNN fixture - HIP engine gradients vs the oracle on the same minibatch."""
import json, os, sys, numpy as np, torch, scipy.sparse as sp
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cvae_oracle as O
from scarches_amd.engine import CVAEEngine
z = np.load(os.path.join(ROOT, 'tests/golden/knn_paired.npz'))
cfg = json.loads(str(z["cfg"]))
n, G, C, Z, H, bs = cfg["n"], cfg["G"], cfg["C"], cfg["Z"], cfg["H"], cfg["batch_size"]
X = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(n, G))
cond = z["cond"]
ocfg = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, dr_rate=cfg["dr_rate"], use_mmd=True, mmd_on="z", beta=1.0, boundary=None)
pre = "s0/"
W0 = OrderedDict((k[len(pre) + 3:], torch.tensor(z[k])) for k in z.files if k.startswith(pre + "w0/"))
tr = z[pre + "train_idx"].astype(np.int64)
Xt = X[tr].tocsr(); Xt.sort_indices()
ct = cond[tr]
samples = z[pre + "samples"].astype(np.int64)
widths = dict(enc0=H[0], enc1=H[1], dec0=H[1], dec1=H[0])
def run(mode, use_mmd=True, use_masks=True, alpha=0.25):
    spec = dict(ocfg, seed=0, use_mmd=use_mmd)
    eng = CVAEEngine(dict(spec), W0, "cuda:0")
    eng.bind_csr(torch.tensor(Xt.indptr.astype(np.int64)).cuda(), torch.tensor(Xt.indices.astype(np.int32)).cuda(), torch.tensor(Xt.data.astype(np.float32)).cuda())
    rows = samples[0, :bs].copy()
    eps = z[pre + "eps_full"][0].copy()
    masks = {k: np.unpackbits(z[pre + "mask_full/" + k][0], axis=1)[:, :widths[k]].copy() for k in widths}
    if mode == "sorted":
        o = np.argsort(ct[rows], kind="stable"); rows = rows[o]; eps = eps[o]; masks = {k: v[o] for k, v in masks.items()}
    if mode == "dedup":
        _, first = np.unique(rows, return_index=True); keep = np.sort(first); rows = rows[keep]; eps = eps[keep]; masks = {k: v[keep] for k, v in masks.items()}
    B = len(rows)
    rt = torch.tensor(rows).cuda()
    dm = {k: torch.tensor(v).cuda().contiguous() for k, v in masks.items()} if use_masks else None
    eng.begin_step(alpha_kl=alpha, lr=1e-3, eps=0.01)
    eng.forward_backward(rt, torch.tensor(ct[rows].astype(np.int32)).cuda(), B, train=True, compute_grad=True, eps=torch.tensor(eps).cuda(), masks=dm if use_masks else None)
    torch.cuda.synchronize()
    xb = torch.tensor(Xt[rows].toarray())
    c2 = dict(ocfg, use_mmd=use_mmd)
    if not use_masks: c2["dr_rate"] = 0.0
    res, grads, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in W0.items()}, c2, xb, ct[rows], xb.sum(1), torch.tensor(eps), alpha, {k: torch.tensor(v) for k, v in masks.items()} if use_masks else None)
    L = eng.losses.cpu().numpy()
    print(mode, "mmd" if use_mmd else "nommd", "masks" if use_masks else "nomasks", "B", B, "losses hip", L[:3], "oracle", res["recon"], res["kl"], res["mmd"])
    worst = 0
    for k in grads:
        g = eng.g[k].cpu().numpy(); r = grads[k].numpy()
        e = np.max(np.abs(g - r)) / max(np.max(np.abs(r)), 1e-30)
        worst = max(worst, e)
        if e > 2e-5: print("    ", k, e)
    print("   worst rel grad err", worst)
    del eng
for mode in ("asis", "sorted", "dedup"):
    run(mode)
run("asis", use_mmd=False)
run("asis", use_masks=False)
run("asis", use_mmd=False, use_masks=False)
