"""Size-independent properties of the HIP train step at the benchmark's full minibatch size (cfg2: 8192 cells x 2000
HVG, 20 conditions, NB, MMD on z, hidden [128, 128], Z = 10) - where the CPU oracle no longer finishes in seconds:

  * two independent implementations agree: fused middle kernels + grouped GEMM vs one GEMM / epilogue launch per layer;
  * launch modes agree: eager launches, the recorded launch plan and the HIP graph give the same trajectory;
  * row-permutation invariance: the step's losses and gradients do not depend on the order of the cells;
  * batch additivity of the likelihood + KL gradients (MMD off): g(B) = (n1 g(B1) + n2 g(B2)) / B;
  * the CSR gather against torch's own sparse->dense path, the MMD (value + gradient) against a plain torch fp64
    evaluation of the reference's pairwise formula (models/trvae/losses.py:166-270).
Tolerances are the north-star ones (rel 1e-5 on losses, rel 1e-5 max-norm per gradient tensor) unless stated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G, C, Z, H, B = 2000, 20, 10, [128, 128], 8192


@pytest.fixture(scope="module")
def data():
    from scarches_amd.synth import make_synthetic_csr
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(40_000, G, C, dev, seed=77)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    rows = torch.randperm(40_000, generator=gen, device=dev)[:B]
    cond = cond_all[rows]
    order = torch.argsort(cond, stable=True)
    return dict(csr=(indptr, indices, vals), rows=rows[order].contiguous(), cond=cond[order].contiguous().to(torch.int32))


def make_engine(data, **over):
    from bench import build_trvae_params
    from scarches_amd.engine import CVAEEngine
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020)
    spec.update(over)
    eng = CVAEEngine(spec, build_trvae_params(spec["G"], spec["C"], spec["H"], spec["Z"]), "cuda:0")
    eng.bind_csr(*data["csr"])
    return eng


def grads(eng):
    return {k: eng.g[k].detach().clone() for k in eng.names}


def assert_grads_close(ga, gb, rtol=1e-5):
    gmax = max(float(v.abs().max()) for v in gb.values())
    for k in gb:
        scale = max(float(gb[k].abs().max()), 2e-6 * gmax)
        err = float((ga[k] - gb[k]).abs().max())
        assert err <= rtol * scale, (k, err, scale)


def test_fused_and_unfused_paths_agree(data):
    a, b = make_engine(data, fused_mid=True), make_engine(data, fused_mid=False)
    assert a.use_fused_mid and not b.use_fused_mid
    eps = torch.randn(B, Z, device="cuda")
    for e in (a, b):
        e.begin_step(alpha_kl=0.5, lr=1e-3)
        e.forward_backward(data["rows"], data["cond"], B, eps=eps)
    torch.cuda.synchronize()
    la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
    assert np.all(np.abs(la[:3] - lb[:3]) <= 1e-5 * np.abs(lb[:3]) + 1e-7), (la, lb)
    assert_grads_close(grads(a), grads(b))


def test_launch_modes_agree(data):
    ref = None
    for mode in (False, "plan", True):
        eng = make_engine(data)
        for _ in range(3):
            eng.train_step(data["rows"], data["cond"], B, 0.5, 1e-3, use_graph=mode)
        torch.cuda.synchronize()
        out = (eng.losses.cpu().numpy().copy(), eng.flat_p.detach().clone())
        if ref is None:
            ref = out
            continue
        # same kernels, same Philox streams; float atomics (theta / bias / MMD columns) may reorder between runs
        assert np.all(np.abs(out[0][:3] - ref[0][:3]) <= 1e-5 * np.abs(ref[0][:3])), (mode, out[0], ref[0])
        assert float((out[1] - ref[1]).abs().max()) <= 1e-5 * float(ref[1].abs().max()), mode


def test_row_permutation_invariance(data):
    eps = torch.randn(B, Z, device="cuda")
    perm = torch.randperm(B, device="cuda")
    a, b = make_engine(data, dr_rate=0.0), make_engine(data, dr_rate=0.0)
    a.begin_step(alpha_kl=0.5, lr=1e-3)
    a.forward_backward(data["rows"], data["cond"], B, eps=eps)
    b.begin_step(alpha_kl=0.5, lr=1e-3)
    b.forward_backward(data["rows"][perm].contiguous(), data["cond"][perm].contiguous(), B, eps=eps[perm].contiguous())
    torch.cuda.synchronize()
    la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
    assert abs(la[0] - lb[0]) <= 1e-5 * abs(la[0]) and abs(la[1] - lb[1]) <= 1e-5 * abs(la[1]), (la, lb)
    # the MMD is a sum of O(10) block means with alternating signs: rel 1e-5 of the largest partial sum
    assert abs(la[2] - lb[2]) <= 1e-4 * abs(la[2]) + 1e-5, (la, lb)
    assert_grads_close(grads(b), grads(a), rtol=1e-4)


def test_batch_additivity_without_mmd(data):
    eps = torch.randn(B, Z, device="cuda")
    n1 = 3000
    full = make_engine(data, use_mmd=False, dr_rate=0.0)
    full.begin_step(alpha_kl=0.5, lr=1e-3)
    full.forward_backward(data["rows"], data["cond"], B, eps=eps)
    gf, lf = grads(full), full.losses.cpu().numpy().copy()
    acc, lsum = None, np.zeros(2)
    for lo, hi in ((0, n1), (n1, B)):
        e = make_engine(data, use_mmd=False, dr_rate=0.0)
        e.begin_step(alpha_kl=0.5, lr=1e-3)
        n = hi - lo
        e.forward_backward(data["rows"][lo:hi].contiguous(), data["cond"][lo:hi].contiguous(), n,
                           eps=eps[lo:hi].contiguous())
        torch.cuda.synchronize()
        w = n / B
        lsum += w * e.losses.cpu().numpy()[:2]
        g = grads(e)
        acc = {k: w * v for k, v in g.items()} if acc is None else {k: acc[k] + w * g[k] for k in g}
    assert np.all(np.abs(lsum - lf[:2]) <= 1e-5 * np.abs(lf[:2])), (lsum, lf)
    # the pieces run their GEMMs with other split-K counts than the full batch: a pre-activation within rounding of 0
    # (about one in 1e7 at these sizes) can land on the other side of the ReLU and move single gradient entries by one
    # cell's contribution (observed 7.6e-5 of the tensor's max for one split choice, 0 for its neighbours)
    assert_grads_close(acc, gf, rtol=3e-4)


def test_gather_matches_torch_sparse(data):
    from scarches_amd import _lib
    lib = _lib.load()
    indptr, indices, vals = data["csr"]
    rows = data["rows"]
    x = torch.empty(B, G, device="cuda")
    sf = torch.empty(B, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.sca_csr_gather_rows(indptr.data_ptr(), indices.data_ptr(), vals.data_ptr(), rows.data_ptr(), B, G, 1,
                                       x.data_ptr(), G, sf.data_ptr(), st), "gather")
    n = indptr.numel() - 1
    dense = torch.sparse_csr_tensor(indptr, indices.to(torch.int64), vals, size=(n, G)).to_dense()[rows]
    torch.cuda.synchronize()
    assert torch.equal(x, torch.log1p(dense))                     # bit-exact: same log1pf on the same floats
    assert float((sf - dense.sum(1)).abs().max()) <= 1e-6 * float(dense.sum(1).max())


def _mmd_torch64(z, cond, n_cond, beta):
    """models/trvae/losses.py:231-270 as written there (condition partition, pairwise mmd_loss_calc), fp64."""
    alphas = torch.tensor([1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1, 5, 10, 15, 20, 25, 30, 35, 100, 1e3, 1e4, 1e5, 1e6],
                          dtype=torch.float32, device=z.device).double()
    bet = 1.0 / (2.0 * alphas)

    def kern(x, y):
        d2 = torch.cdist(x, y).pow(2)
        return torch.exp(-bet.view(-1, 1, 1) * d2.unsqueeze(0)).sum(0)

    parts = [z[cond == c] for c in range(n_cond)]
    loss = z.new_zeros(())
    for i in range(n_cond):
        if parts[i].shape[0] < 1:
            continue
        for j in range(i):
            if parts[j].shape[0] < 1:
                continue
            s, t = parts[i], parts[j]
            loss = loss + kern(s, s).mean() + kern(t, t).mean() - 2 * kern(s, t).mean()
    return beta * loss


def test_mmd_full_size_against_torch_fp64():
    from scarches_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    z = torch.randn(B, Z, device="cuda", generator=gen)
    cond = torch.sort(torch.randint(0, C, (B,), device="cuda", generator=gen))[0].to(torch.int32)
    z64 = z.double().requires_grad_(True)
    ref = _mmd_torch64(z64, cond, C, 1.0)
    ref.backward()
    ns, npart = lib.sca_mmd_num_slabs(B, Z), lib.sca_mmd_num_partials(B, Z)
    Wt = torch.empty(C * C, device="cuda")
    part = torch.zeros(npart, dtype=torch.float64, device="cuda")
    slabs = torch.full((ns, B, Z), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.sca_mmd_fwd_bwd(z.data_ptr(), Z, cond.data_ptr(), B, Z, C, -1, 1.0, Wt.data_ptr(), None,
                                   part.data_ptr(), slabs.data_ptr(), B * Z, st), "mmd")
    torch.cuda.synchronize()
    val, grad = part.sum().item(), slabs.sum(0).double()
    # value: rel 1e-5 of the sum of |block means| (the loss itself is a cancelling sum of C(C-1)/2 * 3 terms of O(10))
    assert abs(val - ref.item()) <= 1e-5 * abs(ref.item()) + 2e-5, (val, ref.item())
    err = float((grad - z64.grad).abs().max())
    assert err <= 5e-5 * float(z64.grad.abs().max()), err


@pytest.mark.parametrize("cfg", [
    dict(Gx=333, Cx=3, Hx=[100, 37], Zx=7, Bx=77, recon="nb", use_ln=True, dr=0.1),
    dict(Gx=512, Cx=5, Hx=[256, 128, 64], Zx=10, Bx=1000, recon="nb", use_ln=True, dr=0.05),      # wide (NS=4) path
    dict(Gx=200, Cx=2, Hx=[64], Zx=3, Bx=33, recon="nb", use_ln=False, dr=0.0),                    # no norm, one layer
    dict(Gx=200, Cx=2, Hx=[64, 32], Zx=5, Bx=130, recon="mse", use_ln=True, dr=0.0),               # MSE head
    dict(Gx=1001, Cx=4, Hx=[128, 128, 128, 128], Zx=16, Bx=4097, recon="nb", use_ln=True, dr=0.2),
])
def test_fused_vs_unfused_odd_shapes(cfg):
    """The fused middle kernels against the one-launch-per-layer path on ragged sizes: widths that are not multiples
    of 4 (scalar load / store paths), wide layers, a single hidden layer, no LayerNorm, batch not a multiple of 32."""
    from bench import build_trvae_params
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    Gx, Cx, Hx, Zx, Bx = cfg["Gx"], cfg["Cx"], cfg["Hx"], cfg["Zx"], cfg["Bx"]
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(6000, Gx, Cx, dev, seed=3)
    rows = torch.arange(Bx, device=dev) * 1 % 6000
    order = torch.argsort(cond_all[rows], stable=True)
    rows = rows[order].contiguous()
    cond = cond_all[rows].to(torch.int32).contiguous()
    # The two paths sum their GEMMs in different orders, so a normalised pre-activation that sits within rounding of 0
    # can take the other ReLU branch in one of them and move single gradient entries by one cell's whole contribution
    # (at 4097 cells x 8 layers x 128 units about one draw in two has such an element).  The gradients are compared on a
    # noise draw where both paths took the same branches everywhere; where they did not, the test checks that every
    # flipped pre-activation is rounding-size and draws again.
    for attempt in range(6):
        eps = torch.randn(Bx, Zx, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + attempt))
        engs = []
        for fused in (True, False):
            spec = dict(G=Gx, C=Cx, H=Hx, Z=Zx, recon=cfg["recon"], use_bn=False, use_ln=cfg["use_ln"], use_mmd=True,
                        mmd_on="z", boundary=None, beta=1.0, dr_rate=cfg["dr"], seed=11, fused_mid=fused)
            P = build_trvae_params(Gx, Cx, Hx, Zx)
            if cfg["recon"] == "mse":      # the reference names the MSE head recon_decoder and has no dispersion
                P = type(P)((k.replace("mean_decoder", "recon_decoder"), v) for k, v in P.items() if k != "theta")
            e = CVAEEngine(spec, P, dev)
            e.bind_csr(indptr, indices, vals)
            e.begin_step(alpha_kl=0.3, lr=1e-3)
            e.forward_backward(rows, cond, Bx, eps=eps)
            engs.append(e)
        torch.cuda.synchronize()
        a, b = engs
        assert a.use_fused_mid and not b.use_fused_mid
        la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
        assert np.all(np.isfinite(la)) and np.all(np.abs(la[:3] - lb[:3]) <= 1e-5 * np.abs(lb[:3]) + 1e-7), (la, lb)
        flipped = 0
        for k in a.ws:
            if k.endswith(".yn") and k in b.ws:
                ya, yb = a.ws[k][:Bx], b.ws[k][:Bx]
                flip = (ya > 0) != (yb > 0)
                if bool(flip.any()):
                    flipped += int(flip.sum())
                    assert float(torch.maximum(ya[flip].abs(), yb[flip].abs()).max()) < 1e-5, k
        if flipped == 0:
            break
    else:
        raise AssertionError("no noise draw without a ReLU gate at rounding distance from 0")
    assert_grads_close(grads(a), grads(b))


@pytest.mark.parametrize("n_small", [6000, 4000, 20])
def test_smaller_minibatch_after_the_full_one(data, n_small):
    """A ragged last minibatch / validation batch on an engine whose workspaces were sized for 8192 cells: the MMD
    kernels use MORE gradient slabs and loss partials at some smaller batch sizes (fewer column chunks per workgroup),
    so the workspaces are sized for the worst case over all sizes - the step must equal that of a fresh engine which
    only ever saw the small batch, and a step at the full size afterwards must be unchanged."""
    L = __import__("scarches_amd._lib", fromlist=["load"]).load()
    sel = torch.sort(torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n_small))
                     [:n_small])[0]
    rows, cond = data["rows"][sel].contiguous(), data["cond"][sel].contiguous()
    eps_big, eps = torch.randn(B, Z, device="cuda"), torch.randn(n_small, Z, device="cuda")
    a, b = make_engine(data, dr_rate=0.0), make_engine(data, dr_rate=0.0)
    a.begin_step(alpha_kl=0.5, lr=1e-3)
    a.forward_backward(data["rows"], data["cond"], B, eps=eps_big)
    torch.cuda.synchronize()
    l_big, g_big = a.losses.cpu().numpy().copy(), grads(a)
    assert L.sca_mmd_num_slabs(n_small, Z) * n_small * Z <= a.ws["mmd_g"].numel()
    assert L.sca_mmd_num_partials(n_small, Z) <= a.ws["mmd_part"].numel()
    for e in (a, b):
        e.begin_step(alpha_kl=0.5, lr=1e-3)
        e.forward_backward(rows, cond, n_small, eps=eps)
    torch.cuda.synchronize()
    la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
    assert np.all(np.abs(la[:3] - lb[:3]) <= 1e-5 * np.abs(lb[:3]) + 1e-7), (la, lb)
    assert_grads_close(grads(a), grads(b))
    a.begin_step(alpha_kl=0.5, lr=1e-3)
    a.forward_backward(data["rows"], data["cond"], B, eps=eps_big)
    torch.cuda.synchronize()
    assert np.all(np.abs(a.losses.cpu().numpy()[:3] - l_big[:3]) <= 1e-6 * np.abs(l_big[:3]))
    assert_grads_close(grads(a), g_big, rtol=1e-5)


def test_split_k_slabs_of_a_smaller_minibatch_fit(data):
    """The split-K count is not monotone in the batch size: with G = 2000 and a 256-wide first layer a 960-row
    minibatch takes S = 32 slabs where the 1024-row one takes 16, i.e. 1.9x the slab floats.  The slab workspaces are
    sized for the worst case over every B' <= B; a canary behind each stays intact, and the 960-row step equals that
    of a fresh engine that only ever saw 960 rows."""
    L = __import__("scarches_amd._lib", fromlist=["load"]).load()
    Hh = [256, 64]
    a, b = make_engine(data, H=Hh, dr_rate=0.0, fused_mid=False), make_engine(data, H=Hh, dr_rate=0.0, fused_mid=False)
    n_big, n_small = 1024, 960
    assert (L.sca_gemm_splitk_heuristic(n_small, 256, G) * n_small * 256
            > L.sca_gemm_splitk_heuristic(n_big, 256, G) * n_big * 256)        # the shape that used to overflow
    a._ensure(n_big)
    need = a.ws["slab"].numel()
    assert need >= L.sca_gemm_splitk_heuristic(n_small, 256, G) * n_small * 256
    # canaries: re-home both slab workspaces inside larger allocations whose tails must stay untouched
    for key in ("slab", "slab_aux"):
        big = torch.full((need + 4096,), 123.0, device="cuda")
        a.ws[key] = big[:need]
        a.ws[key + "_canary"] = big
    a._invalidate()
    for n in (n_big, n_small):
        rows, cond = data["rows"][:n].contiguous(), data["cond"][:n].contiguous()
        eps = torch.randn(n, Z, device="cuda")
        for e in ((a, b) if n == n_small else (a,)):
            e.begin_step(alpha_kl=0.5, lr=1e-3)
            e.forward_backward(rows, cond, n, eps=eps)
        torch.cuda.synchronize()
        for key in ("slab", "slab_aux"):
            assert bool((a.ws[key + "_canary"][need:] == 123.0).all()), (key, n)
    la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
    assert np.all(np.abs(la[:3] - lb[:3]) <= 1e-5 * np.abs(lb[:3]) + 1e-7), (la, lb)
    assert_grads_close(grads(a), grads(b))
    with pytest.raises(RuntimeError):       # a GEMM whose slabs do not fit is refused, never launched
        a.ws["slab"] = a.ws["slab"][:1024]
        a._gemm(0, 0, a.ws["x"].data_ptr(), G, a.p["encoder.FC.L0.expr_L.weight"].data_ptr(), G, n_small, 256, G, 0)


@pytest.mark.parametrize("mode", ["plan", True])
def test_recorded_plans_survive_a_workspace_regrowth(data, mode):
    """train -> encode more cells than the training batch (the workspaces are reallocated) -> train again: recorded
    launch plans / HIP graphs hold raw workspace addresses and must be dropped with the old buffers.  The trajectory
    equals that of an engine that runs the same steps eagerly."""
    n, n_enc = 512, 3000
    rows, cond = data["rows"][:n].contiguous(), data["cond"][:n].contiguous()
    engs = [make_engine(data), make_engine(data)]
    for eng, m in zip(engs, (mode, False)):
        for _ in range(3):
            eng.train_step(rows, cond, n, 0.5, 1e-3, use_graph=m)
        gen0 = eng._ws_gen
        z, mu, _ = eng.encode(data["rows"][:n_enc].contiguous(), data["cond"][:n_enc].contiguous(), n_enc)
        assert eng._ws_gen > gen0 and not eng._plans
        junk = [torch.full((n_enc, G), float("nan"), device="cuda") for _ in range(4)]   # reuse freed blocks
        for _ in range(3):
            eng.train_step(rows, cond, n, 0.5, 1e-3, use_graph=m)
        torch.cuda.synchronize()
        del junk
    a, b = engs
    assert torch.isfinite(a.flat_p).all()
    la, lb = a.losses.cpu().numpy(), b.losses.cpu().numpy()
    assert np.all(np.abs(la[:3] - lb[:3]) <= 2e-5 * np.abs(lb[:3]) + 1e-7), (la, lb)
    d = float((a.flat_p - b.flat_p).abs().max())
    assert d <= 1e-5 * float(b.flat_p.abs().max()), d


@pytest.mark.parametrize("Hx,Zx,Bx,mask", [([128, 128], 10, 1000, "3"), ([128, 128, 128], 16, 77, "3"),
                                           ([256, 64], 10, 515, "7"), ([256, 64], 10, 515, "3"), ([64], 5, 130, "3"),
                                           ([96, 32, 160], 7, 300, "3"), ([96, 32, 160], 7, 300, "7")])
def test_mid_direct_family_matches_staged_family(Hx, Zx, Bx, mask, monkeypatch):
    """sca_mid_fwd / sca_mid_bwd (fragments straight from L2: the straight-line 128-wide kernels, the generic direct
    kernels) against sca_mid_fwd_staged / sca_mid_bwd_staged (weights through LDS) on the SAME argument block: every
    tensor either pass writes agrees to rounding (the two families sum in different orders).  SCA_MID_DIRECT=3 (the
    default) takes the straight-line kernels - 128-wide, or the generalised ones for widths that are multiples of 32 -
    and 7 forces the generic loop-form kernels for every shape but the 128-wide one."""
    monkeypatch.setenv("SCA_MID_DIRECT", mask)
    from bench import build_trvae_params
    from scarches_amd import _lib
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    Gx, Cx = 300, 3
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(3000, Gx, Cx, dev, seed=5)
    rows = torch.arange(Bx, device=dev)
    order = torch.argsort(cond_all[rows], stable=True)
    rows = rows[order].contiguous()
    cond = cond_all[rows].to(torch.int32).contiguous()
    eps = torch.randn(Bx, Zx, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    spec = dict(G=Gx, C=Cx, H=Hx, Z=Zx, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.1, seed=11, fused_mid=True)
    e = CVAEEngine(spec, build_trvae_params(Gx, Cx, Hx, Zx), dev)
    e.bind_csr(indptr, indices, vals)
    e.begin_step(alpha_kl=0.3, lr=1e-3)
    e.forward_backward(rows, cond, Bx, eps=eps)
    torch.cuda.synchronize()
    L = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    (_, rf), = [v for k, v in e._mid_cache.items() if k[7] is None]
    (_, rb), = [v for k, v in e._mid_cache.items() if k[7] is not None]
    fwd_keys = [k for k in e.ws if k.endswith((".yn", ".h", ".rs")) or k in ("z", "mu_lv", "kl_row")]
    bwd_keys = [k for k in e.ws if k.endswith(".da")] + ["dheads"]
    bwd_keys = [k for k in bwd_keys if k in e.ws]
    assert fwd_keys and bwd_keys

    def run(fn, ref, keys):
        for k in keys:
            e.ws[k].fill_(float("nan")) if e.ws[k].dtype == torch.float32 else None
        _lib.check(fn(ref, st), "mid")
        torch.cuda.synchronize()
        return {k: e.ws[k][:Bx].clone() for k in keys}
    # forward first (the backward pass reads what it saved); the staged forward runs last so that both backward passes
    # read the same saved tensors
    fa = run(L.sca_mid_fwd, rf, fwd_keys)
    fb = run(L.sca_mid_fwd_staged, rf, fwd_keys)
    ba = run(L.sca_mid_bwd, rb, bwd_keys)
    bb = run(L.sca_mid_bwd_staged, rb, bwd_keys)
    for got, want in ((fa, fb), (ba, bb)):
        for k in want:
            w = want[k]
            assert torch.isfinite(w).all(), k
            scale = float(w.abs().max())
            err = float((got[k] - w).abs().max())
            assert err <= 1e-5 * scale + 1e-7, (k, err, scale)


def test_captured_step_with_side_streams_waiting_for_each_other(data, monkeypatch):
    """hipStreamEndCapture of this ROCm segfaults when two forked streams of one capture wait for each other in both
    directions (scripts/dbg/capture_crash_repro.py).  SCA_OUT_DW_AFTER_MMD=1 makes the step contain exactly that (the
    output layer's weight-gradient GEMM on the second side stream waits for the MMD on the first, the gradient tail on
    the first later waits for that GEMM): the engine routes the second direction through the origin stream while a
    graph is captured, and the captured step gives the trajectory of the plain one."""
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SCA_OUT_DW_AFTER_MMD", flag)
        eng = make_engine(data)
        for _ in range(3):
            eng.train_step(data["rows"], data["cond"], B, 0.5, 1e-3, use_graph=True)
        torch.cuda.synchronize()
        outs.append((eng.losses.cpu().numpy().copy(), eng.flat_p.detach().clone()))
    (la, pa), (lb, pb) = outs
    assert np.all(np.abs(la[:3] - lb[:3]) <= 1e-5 * np.abs(lb[:3])), (la, lb)
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pb.abs().max())


@pytest.mark.parametrize("Hx,Zx,Bx", [([128, 128], 10, 1000), ([256, 64], 10, 515), ([100, 37], 7, 77), ([64], 5, 130)])
@pytest.mark.parametrize("mean", [True, False])
def test_encode_fused_matches_per_layer_encode(Hx, Zx, Bx, mean, monkeypatch):
    """get_latent's encoder pass: first-layer GEMM + ONE sca_mid_fwd launch with n_dec = 0 (straight-line 128-wide,
    generalised, staged kernels by shape) against the GEMM + epilogue pair per layer, eval mode, mean and sampled z
    (injected noise)."""
    from bench import build_trvae_params
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    Gx, Cx = 300, 3
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(3000, Gx, Cx, dev, seed=5)
    rows = torch.arange(Bx, device=dev)
    cond = cond_all[rows].to(torch.int32).contiguous()
    eps = torch.randn(Bx, Zx, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SCA_ENCODE_FUSED", flag)
        spec = dict(G=Gx, C=Cx, H=Hx, Z=Zx, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z",
                    boundary=None, beta=1.0, dr_rate=0.1, seed=11)
        e = CVAEEngine(spec, build_trvae_params(Gx, Cx, Hx, Zx), dev)
        e.bind_csr(indptr, indices, vals)
        z, mu, var = e.encode(rows, cond, Bx, mean=mean, eps=eps)
        torch.cuda.synchronize()
        outs.append((z.clone(), mu.clone(), var.clone()))
    for a, b in zip(*outs):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7
    if mean:
        assert torch.equal(outs[0][0], outs[0][1])          # z is the mean
    else:
        assert float((outs[0][0] - outs[0][1]).abs().max()) > 1e-3
