"""Debug: replay of the paired kNN fixture, HIP engine and oracle in lockstep; parameter distance after every step."""
import json, os, sys, numpy as np, torch, scipy.sparse as sp
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cvae_oracle as O
from scarches_amd.engine import CVAEEngine
torch.set_num_threads(8)
z = np.load(os.path.join(ROOT, 'tests/golden/knn_paired.npz'))
cfg = json.loads(str(z["cfg"]))
n, G, C, Z, H, bs = cfg["n"], cfg["G"], cfg["C"], cfg["Z"], cfg["H"], cfg["batch_size"]
X = sp.csr_matrix((z["x_data"], z["x_indices"], z["x_indptr"]), shape=(n, G))
cond = z["cond"]
ocfg = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, dr_rate=cfg["dr_rate"], use_mmd=True, mmd_on="z", beta=1.0, boundary=None)
pre = "s%s/" % os.environ.get("DBG_SEED", "0")
W0 = OrderedDict((k[len(pre) + 3:], torch.tensor(z[k])) for k in z.files if k.startswith(pre + "w0/"))
tr = z[pre + "train_idx"].astype(np.int64)
Xt = X[tr].tocsr(); Xt.sort_indices()
Xd = torch.tensor(Xt.toarray())
ct = cond[tr]
samples = z[pre + "samples"].astype(np.int64)
widths = dict(enc0=H[0], enc1=H[1], dec0=H[1], dec1=H[0])
eng = CVAEEngine(dict(ocfg, seed=0), W0, "cuda:0")
eng.bind_csr(torch.tensor(Xt.indptr.astype(np.int64)).cuda(), torch.tensor(Xt.indices.astype(np.int32)).cuda(), torch.tensor(Xt.data.astype(np.float32)).cuda())
eng.overlap = os.environ.get("DBG_OVERLAP", "1") == "1"
P = {k: v.clone() for k, v in W0.items()}
M = {k: torch.zeros_like(v) for k, v in P.items()}; V = {k: torch.zeros_like(v) for k, v in P.items()}
n_full = samples.shape[1] // bs; tail = samples.shape[1] - n_full * bs
i_full = i_tail = 0; t = 0; KEEP = []
for epoch in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    alpha = min(cfg["alpha_kl"] * epoch / cfg["alpha_epoch_anneal"], cfg["alpha_kl"])
    for it in range(n_full + (1 if tail else 0)):
        rows = samples[epoch, it * bs:(it + 1) * bs]; B = len(rows); full = B == bs
        eps = z[pre + ("eps_full" if full else "eps_tail")][i_full if full else i_tail]
        masks = {k: np.unpackbits(z[pre + ("mask_full/" if full else "mask_tail/") + k][i_full if full else i_tail], axis=1)[:, :widths[k]].copy() for k in widths}
        if full: i_full += 1
        else: i_tail += 1
        eng.begin_step(alpha_kl=alpha, lr=cfg["lr"], eps=cfg["eps"])
        tmp = (torch.tensor(rows).cuda(), torch.tensor(ct[rows].astype(np.int32)).cuda(), torch.tensor(eps).cuda(), {k: torch.tensor(v).cuda().contiguous() for k, v in masks.items()})
        if os.environ.get("DBG_KEEP", "0") == "1": KEEP.append(tmp)
        eng.forward_backward(tmp[0], tmp[1], B, train=True, compute_grad=True, eps=tmp[2], masks=tmp[3])
        del tmp
        MODE = os.environ.get("DBG_SYNC", "both")
        if MODE in ("both", "mid"): torch.cuda.synchronize()
        gh = {k: eng.g[k].cpu().clone() for k in P} if MODE == "both" else {k: torch.zeros_like(P[k]) for k in P}
        eng.adam_step(weight_decay=cfg["weight_decay"])
        if MODE in ("both", "end"): torch.cuda.synchronize()
        xb = Xd[rows]
        res, grads, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in P.items()}, ocfg, xb, ct[rows], xb.sum(1), torch.tensor(eps), alpha, {k: torch.tensor(v) for k, v in masks.items()})
        t += 1
        for k in P:
            O.adam_step(P[k], grads[k], M[k], V[k], t, lr=cfg["lr"], eps=cfg["eps"], wd=cfg["weight_decay"])
        ge = max((float((gh[k] - grads[k]).abs().max() / max(float(grads[k].abs().max()), 1e-30)), k) for k in P)
        if t % 15 and MODE != "both": continue
        pe = max((float((eng.p[k].cpu() - P[k]).abs().max() / float(P[k].abs().max())), k) for k in P)
        L = eng.losses.cpu().numpy()
        if t == int(os.environ.get('DBG_STEP', '21')):
            for k in P:
                d = (gh[k] - grads[k]).abs(); sc = float(grads[k].abs().max())
                if d.dim() == 2:
                    rows_ = (d.max(1).values / sc); print('   ', k, 'rel err %.2e' % float(d.max() / sc), 'rows with err>1e-4:', (rows_ > 1e-4).nonzero().flatten().tolist()[:12])
                else:
                    print('   ', k, 'rel err %.2e' % float(d.max() / sc), 'idx', (d / sc > 1e-4).nonzero().flatten().tolist()[:12])
            for tag in ("enc0", "enc1", "dec0", "dec1"):
                yn = eng.ws[tag + ".yn"][:B].cpu()
                a = yn.abs()
                i = int(a.argmin())
                print("   step", t, tag, "min |yn| %.3e" % float(a.min()), "value %.9e" % float(yn.flatten()[i]), "at", i // yn.shape[1], i % yn.shape[1], "n(|yn|<1e-6)", int((a < 1e-6).sum()))
        jump = pe[0] > 3 * globals().get("LASTPE", 1e-9) and pe[0] > 2e-5
        globals()["LASTPE"] = max(pe[0], 1e-7)
        if os.environ.get("DBG_JUMPS") and not (jump or t % 30 == 0): continue
        print("JUMP" if jump else "    ", "step", t, "B", B, "alpha", alpha, "grad err", "%.2e %s" % ge, "param err", "%.2e %s" % pe, "recon %.4f/%.4f mmd %.5f/%.5f" % (L[0], res["recon"], L[2], res["mmd"]), flush=True)
