"""The data-parallel step path (`train_step(world > 1)`: forward/backward plan or graph, RCCL all-reduce of the flat
gradient buffer on the launch stream, then the Adam plan or graph) exercised on ONE GPU with a one-rank nccl group:
the all-reduce is then the identity and `world=2` halves the gradient in every launch mode alike, so eager, recorded
plan and HIP graph must give the same trajectory.  (Multi-rank semantics are covered by the gloo tests on CPU.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G, C, Z, H = 2000, 20, 10, [128, 128]


@pytest.fixture(scope="module")
def group():
    import torch.distributed as dist
    if dist.is_initialized():
        yield
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [512, 4096])
def test_dp_step_modes_agree(group, B):
    from bench import build_trvae_params
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    dev = torch.device("cuda", 0)
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(20_000, G, C, dev, seed=5)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    rows = torch.randperm(20_000, generator=gen, device=dev)[:B]
    order = torch.argsort(cond_all[rows], stable=True)
    rows = rows[order].contiguous()
    cond = cond_all[rows].to(torch.int32).contiguous()
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020)
    ref = None
    for mode in (False, "plan", True):
        eng = CVAEEngine(spec, build_trvae_params(G, C, H, Z), dev)
        eng.bind_csr(indptr, indices, vals)
        for _ in range(3):
            eng.train_step(rows, cond, B, 0.5, 1e-3, world=2, use_graph=mode, grad_weight=0.5)
        torch.cuda.synchronize()
        out = (eng.losses.cpu().numpy().copy(), eng.flat_p.detach().clone())
        assert np.all(np.isfinite(out[0]))
        if ref is None:
            ref = out
            continue
        # three optimiser steps: float-atomic ordering noise (theta / bias / MMD column sums) compounds, rel 1e-4
        assert np.all(np.abs(out[0][:3] - ref[0][:3]) <= 1e-4 * np.abs(ref[0][:3])), (mode, out[0], ref[0])
        assert float((out[1] - ref[1]).abs().max()) <= 1e-4 * float(ref[1].abs().max()), mode


def test_epoch_log_weights_of_ragged_slices_add_up():
    """Epoch logs under data parallelism: a rank accumulates its minibatch means with weight n_local / n_global
    (hyper[6] of sca_finalize_losses, Trainer._set_log_weight).  Two 'ranks' on one GPU over ragged slices of three
    global minibatches: the summed accumulators equal those of the single-device run over the whole minibatches."""
    from bench import build_trvae_params
    from scarches_amd import distributed as sdist
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    dev = torch.device("cuda", 0)
    Gs, Cs, Hs, Zs = 300, 4, [32, 16], 6
    indptr, indices, vals, cond_all, _ = make_synthetic_csr(3000, Gs, Cs, dev, seed=8)
    spec = dict(G=Gs, C=Cs, H=Hs, Z=Zs, recon="nb", use_bn=False, use_ln=True, use_mmd=False, dr_rate=0.0, seed=1)

    def engine():
        e = CVAEEngine(spec, build_trvae_params(Gs, Cs, Hs, Zs), dev)
        e.bind_csr(indptr, indices, vals)
        e.set_hyper(alpha_kl=0.5)
        return e
    gen = torch.Generator(device=dev).manual_seed(3)
    batches = [torch.randperm(3000, generator=gen, device=dev)[:n] for n in (101, 101, 37)]
    eps = [torch.randn(b.numel(), Zs, device=dev, generator=gen) for b in batches]
    full, ranks = engine(), [engine(), engine()]
    for b, e_ in zip(batches, eps):
        full.forward_backward(b, cond_all[b].to(torch.int32), b.numel(), compute_grad=False, eps=e_)
        for r, eng in enumerate(ranks):
            lo, hi = sdist.shard_range(b.numel(), r, 2)
            eng.set_hyper(log_weight=(hi - lo) / b.numel())
            eng.forward_backward(b[lo:hi].contiguous(), cond_all[b[lo:hi]].to(torch.int32), hi - lo,
                                 compute_grad=False, eps=e_[lo:hi].contiguous())
    torch.cuda.synchronize()
    ref = full.accum.cpu().numpy()
    got = ranks[0].accum.cpu().numpy() + ranks[1].accum.cpu().numpy()
    assert abs(ref[4] - 3.0) < 1e-12 and abs(got[4] - 3.0) < 1e-6
    assert np.all(np.abs(got[:2] - ref[:2]) <= 1e-5 * np.abs(ref[:2])), (got, ref)
