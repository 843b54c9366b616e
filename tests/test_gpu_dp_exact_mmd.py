"""Data-parallel MMD over the GLOBAL minibatch (CVAEEngine.set_dp(exact_mmd=True), SURVEY 8(e) exception 1).

The reference's MMD (scarches/models/trvae/losses.py:231-270) couples every pair of cells of a minibatch.  Under DP each
rank all-gathers the latent rows + conditions and evaluates ITS ROW BLOCK against all columns (sca_mmd_rows_fwd_bwd);
value and gradient rows carry n_global / n_local, so that the n_local / n_global-weighted all-reduces (gradients per
step, log sums per epoch) reproduce the single-device MMD and its gradient.  Checked here on ONE GPU: three "ranks" run
one after the other on ragged slices (334 / 334 / 333 cells, so padding rows with condition -1 are in play), the
all-gather is replaced by a function that assembles what the other ranks would have sent (from a single-device run of
the whole minibatch), and   sum_r (n_r / n) * grad_r   must equal the single-device gradient of the whole minibatch,
sum_r (n_r / n) * mmd_r the single-device MMD value (like the recon and KL means).  Both launch paths (fused middle kernels, and MMD on the
first decoder layer 'y' through the per-layer path) are covered.  The collective itself
(torch.distributed.all_gather_into_tensor) runs for real in tests/test_gpu_bench_dp.py (two gloo ranks on one GPU) and
tests/test_dp_gloo.py (CPU)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_fullsize import assert_grads_close, grads

pytestmark = pytest.mark.gpu

G, C, Z, H = 512, 6, 10, [64, 32]
WORLD, NG = 3, 1001


@pytest.fixture(scope="module")
def data():
    from scarches_amd import distributed as sdist
    from scarches_amd.synth import make_synthetic_csr
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(6000, G, C, dev, seed=3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    batch = torch.randperm(6000, generator=gen, device=dev)[:NG]
    slices = []
    for r in range(WORLD):
        b = sdist.local_slice(batch, r, WORLD)
        c = cond_all[b]
        order = torch.argsort(c, stable=True)
        slices.append((b[order].contiguous(), c[order].contiguous().to(torch.int32)))
    eps = torch.randn(NG, Z, device=dev, generator=gen)
    return dict(csr=(indptr, indices, vals), slices=slices, eps=eps)


def make_engine(data, **over):
    from bench import build_trvae_params
    from scarches_amd.engine import CVAEEngine
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.0, seed=2020)
    spec.update(over)
    eng = CVAEEngine(spec, build_trvae_params(G, C, H, Z), "cuda:0")
    eng.bind_csr(*data["csr"])
    return eng


@pytest.mark.parametrize("mmd_on,boundary", [("z", None), ("y", None), ("z", 2)])
def test_exact_dp_mmd_matches_single_device(data, mmd_on, boundary):
    sizes = [s[0].numel() for s in data["slices"]]
    assert sizes == [334, 334, 333]
    n_max = max(sizes)
    rows_full = torch.cat([s[0] for s in data["slices"]])
    cond_full = torch.cat([s[1] for s in data["slices"]])
    # ---- single device, whole minibatch
    full = make_engine(data, mmd_on=mmd_on, boundary=boundary)
    full.begin_step(alpha_kl=0.5, lr=1e-3)
    full.forward_backward(rows_full, cond_full, NG, eps=data["eps"])
    torch.cuda.synchronize()
    gf, lf = grads(full), full.losses.cpu().numpy().copy()
    Dm = Z if mmd_on == "z" else H[-1]
    feat_full = (full.ws["z"] if mmd_on == "z" else full.ws["dec0.h"]).view(-1)[:NG * Dm].view(NG, Dm).clone()
    assert lf[2] > 0
    # what an all-gather of the padded per-rank blocks delivers
    feat_all = torch.zeros(WORLD * n_max, Dm, device="cuda")
    cond_all = torch.full((WORLD * n_max,), -1, dtype=torch.int32, device="cuda")
    lo = 0
    for r, n in enumerate(sizes):
        feat_all[r * n_max:r * n_max + n] = feat_full[lo:lo + n]
        cond_all[r * n_max:r * n_max + n] = cond_full[lo:lo + n]
        lo += n

    acc, lsum, lo = None, np.zeros(3), 0
    for r, (rows, cond) in enumerate(data["slices"]):
        n = rows.numel()
        calls = []

        def gather(out, local, r=r, n=n, calls=calls):
            exp = cond_all if local.dtype == torch.int32 else feat_all
            mine = exp[r * n_max:(r + 1) * n_max]
            if local.dtype == torch.int32:
                assert torch.equal(local, mine)
            else:       # this rank's rows were recomputed on a smaller batch (other split-K counts): rounding only
                assert float((local - mine).abs().max()) <= 1e-5 * float(mine.abs().max())
                assert float(local[n:].abs().sum()) == 0.0
            out.copy_(exp)
            calls.append(local.dtype)

        e = make_engine(data, mmd_on=mmd_on, boundary=boundary)
        e.set_dp(r, WORLD, exact_mmd=True, gather=gather)
        eps_r = data["eps"][lo:lo + n].contiguous()
        e.begin_step(alpha_kl=0.5, lr=1e-3)
        e.forward_backward(rows, cond, n, eps=eps_r, n_global=NG)
        assert len(calls) == 2
        torch.cuda.synchronize()
        le = e.losses.cpu().numpy()
        w = n / NG
        lsum += w * le[:3]
        g = grads(e)
        acc = {k: w * v for k, v in g.items()} if acc is None else {k: acc[k] + w * g[k] for k in g}
        lo += n
    assert np.all(np.abs(lsum - lf[:3]) <= 1e-5 * np.abs(lf[:3]) + 1e-6), (lsum, lf)
    assert_grads_close(acc, gf, rtol=2e-4)


def test_exact_dp_train_step_issues_one_gather_pair_per_step_in_every_launch_mode(data):
    """train_step under exact DP MMD: eager launches and the recorded launch plan (a requested HIP graph becomes a plan:
    the gathers are host calls) issue exactly one gather pair per step - the plan's warm-up and recording passes gather
    nothing, so ranks may record at different steps - and all modes give the same trajectory."""
    import torch.distributed as dist
    if not dist.is_initialized():
        import os
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    rows, cond = data["slices"][2]
    n = rows.numel()                    # 333: n * WORLD = 999 is an even split
    n_max = n
    other = torch.randn(WORLD * (n_max + 1), Z, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    ocond = (torch.arange(WORLD * (n_max + 1), device="cuda") % C).to(torch.int32)

    def run(mode, n_global):
        calls = []

        def gather(out, local):
            if local.dtype == torch.int32:
                out.copy_(ocond[:out.numel()])
            else:
                out.copy_(other[:out.shape[0]])
            m = local.shape[0]
            out[2 * m:2 * m + m] = local        # rank 2's own block
            calls.append(1)
        e = make_engine(data)
        e.set_dp(2, WORLD, exact_mmd=True, gather=gather)
        counts = []
        for _ in range(3):
            e.train_step(rows, cond, n, 0.5, 1e-3, world=WORLD, use_graph=mode, grad_weight=n / n_global,
                         n_global=n_global)
            counts.append(len(calls))
        torch.cuda.synchronize()
        return counts, e.losses.cpu().numpy().copy(), e.flat_p.detach().clone()

    c_eager, l_eager, p_eager = run(False, n * WORLD)
    assert c_eager == [2, 4, 6]
    for mode in ("plan", True, "auto"):
        c, l, p = run(mode, n * WORLD)
        assert c == [2, 4, 6], c
        assert np.all(np.abs(l[:3] - l_eager[:3]) <= 1e-4 * np.abs(l_eager[:3])), (mode, l, l_eager)
        assert float((p - p_eager).abs().max()) <= 1e-4 * float(p_eager.abs().max())
    c, _, _ = run("plan", n * WORLD + 1)   # ragged
    assert c == [2, 4, 6]


def test_per_rank_mmd_differs_and_is_selectable(data):
    """exact_mmd=False keeps the per-rank estimator: no gather is issued and the MMD value is that of the slice."""
    rows, cond = data["slices"][0]
    n = rows.numel()
    a = make_engine(data)

    def boom(out, local):
        raise AssertionError("per-rank MMD must not gather")
    a.set_dp(0, WORLD, exact_mmd=False, gather=boom)
    b = make_engine(data)
    for e in (a, b):
        e.begin_step(alpha_kl=0.5, lr=1e-3)
        e.forward_backward(rows, cond, n, eps=data["eps"][:n].contiguous(), n_global=NG)
    torch.cuda.synchronize()
    assert np.allclose(a.losses.cpu().numpy(), b.losses.cpu().numpy(), rtol=1e-6)
