"""Data-parallel host logic on CPU: two `gloo` ranks (the GPU path runs the same code over RCCL).

* every rank draws the same global sample and keeps a disjoint slice that covers the minibatch;
* the weighted gradient all-reduce reproduces the single-process global-minibatch gradient (checked with the oracle's
  autograd gradients as the per-rank inputs, ragged shards included);
* the exact-MMD recipe of CVAEEngine.set_dp (padded all-gather of latent rows, full MMD per rank, own rows scaled by
  n_global / n_local) reproduces the single-process gradient including the MMD term;
* summed epoch logs give every rank the same early-stopping / LR decision.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._util import dense_from_case, load_case

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from oracle import cvae_oracle as O
    from scarches_amd import distributed as sdist
    from scarches_amd.utils.monitor import EarlyStopping
    res = {}
    assert sdist.is_dist()

    # ---- (1) same global sample on every rank, disjoint covering slices
    gen = torch.Generator().manual_seed(2020)
    w = torch.rand(1000, generator=torch.Generator().manual_seed(1)).double()
    ids = torch.multinomial(w, 257, replacement=True, generator=gen)
    mine = sdist.local_slice(ids, rank, WORLD)
    gathered = [torch.zeros(129 if r == 0 else 128, dtype=ids.dtype) for r in range(WORLD)]
    dist.all_gather_object(obj := [None] * WORLD, mine.tolist())
    res["cover"] = torch.equal(torch.tensor(sum(obj, [])), ids)
    del gathered

    # ---- (2) weighted all-reduce == global-minibatch gradient (ragged: 15 cells -> 8 + 7)
    z, cfg, W, _, _ = load_case("trvae_nb_ln_small")
    cfg = dict(cfg, use_mmd=False)          # MMD couples cells across ranks (per-rank estimator)
    x = dense_from_case(z, cfg["G"])[:15]
    cond, sf, eps = z["batch"][:15], torch.tensor(z["sizefactor"][:15]), torch.tensor(z["eps"][:15])
    alpha = float(z["alpha"])
    _, g_glob, _ = O.trvae_loss_and_grads(W, cfg, x, cond, sf, eps, alpha)
    lo, hi = sdist.shard_range(15, rank, WORLD)
    _, g_loc, _ = O.trvae_loss_and_grads(W, cfg, x[lo:hi], cond[lo:hi], sf[lo:hi], eps[lo:hi], alpha)
    names = sorted(g_glob)
    flat = torch.cat([g_loc[k].reshape(-1) for k in names])
    sdist.allreduce_mean_(flat, WORLD, weight=(hi - lo) / 15)
    ref = torch.cat([g_glob[k].reshape(-1) for k in names])
    res["grad_err"] = float((flat - ref).abs().max() / ref.abs().max())
    # equal shards: plain mean
    _, g_glob, _ = O.trvae_loss_and_grads(W, cfg, x[:14], cond[:14], sf[:14], eps[:14], alpha)
    lo, hi = sdist.shard_range(14, rank, WORLD)
    _, g_loc, _ = O.trvae_loss_and_grads(W, cfg, x[lo:hi], cond[lo:hi], sf[lo:hi], eps[lo:hi], alpha)
    flat = torch.cat([g_loc[k].reshape(-1) for k in names])
    sdist.allreduce_mean_(flat, WORLD)
    ref = torch.cat([g_glob[k].reshape(-1) for k in names])
    res["grad_err_equal"] = float((flat - ref).abs().max() / ref.abs().max())

    # ---- (2b) exact MMD under DP (CVAEEngine.set_dp): all-gather of the padded latent rows + conditions (padding
    # rows carry condition -1), full MMD on every rank, own rows differentiable, scaled by n_global / n_local so the
    # weighted all-reduce adds it with weight 1  ==  the single-process gradient WITH the MMD term (ragged 8 + 7)
    cfg_m = dict(cfg, use_mmd=True, mmd_on="z")
    res_glob, g_glob, _ = O.trvae_loss_and_grads(W, cfg_m, x, cond, sf, eps, alpha)
    lo, hi = sdist.shard_range(15, rank, WORLD)
    n_loc, n_max = hi - lo, -(-15 // WORLD)
    P = {k: torch.tensor(np.asarray(v)).requires_grad_(True) for k, v in W.items()}
    recon, kl, _, aux = O.trvae_forward(P, dict(cfg_m, use_mmd=False), x[lo:hi].float(),
                                        torch.as_tensor(cond[lo:hi]).long(), sf[lo:hi].float(), eps[lo:hi].float())
    zpad = torch.zeros(n_max, cfg["Z"])
    cpad = torch.full((n_max,), -1, dtype=torch.int32)
    zpad[:n_loc] = aux["z"].detach()
    cpad[:n_loc] = torch.as_tensor(cond[lo:hi]).int()
    zall, call = torch.empty(WORLD * n_max, cfg["Z"]), torch.empty(WORLD * n_max, dtype=torch.int32)
    dist.all_gather_into_tensor(zall, zpad)
    dist.all_gather_into_tensor(call, cpad)
    blocks = [aux["z"] if r == rank else zall[r * n_max:(r + 1) * n_max] for r in range(WORLD)]
    blocks[rank] = torch.cat([aux["z"], torch.zeros(n_max - n_loc, cfg["Z"])])
    zcat, keep = torch.cat(blocks), call >= 0
    mmd = O.mmd_loss(zcat[keep], call[keep].long(), cfg["C"], cfg["beta"], cfg["boundary"])
    (recon + alpha * kl + (15 / n_loc) * mmd).backward()
    flat = torch.cat([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).reshape(-1) for k in names])
    sdist.allreduce_mean_(flat, WORLD, weight=n_loc / 15)
    ref = torch.cat([g_glob[k].reshape(-1) for k in names])
    res["grad_err_mmd"] = float((flat - ref).abs().max() / ref.abs().max())
    res["mmd_err"] = abs(float(mmd) - res_glob["mmd"]) / abs(res_glob["mmd"])

    # ---- (3) summed logs -> identical early-stopping decisions on every rank
    rng = np.random.default_rng(100 + rank)          # different local validation losses
    es = EarlyStopping(patience=4, lr_patience=2)
    decisions = []
    for epoch in range(12):
        acc = torch.tensor([5.0 - 0.3 * min(epoch, 4) + 0.05 * rng.random(), 1.0], dtype=torch.float64)
        sdist.allreduce_sum_(acc)
        metric = float(acc[0] / acc[1])
        es.update_state(metric)
        cont, lr = es.step(metric)
        decisions.append((bool(cont), bool(lr)))
        if not cont:
            break
    # ---- (4) epoch logs on ragged shards: every rank accumulates its minibatch means with weight n_local / n_global
    # (hyper[6] of sca_finalize_losses, Trainer._set_log_weight); the all-reduced sums are the single-process epoch
    # means over the GLOBAL minibatches (sizes 15, 15, 7 -> slices 8+7, 8+7, 4+3)
    acc = torch.zeros(3, dtype=torch.float64)           # [sum w*recon, sum w*kl, sum w]
    ref = torch.zeros(3, dtype=torch.float64)
    xx = dense_from_case(z, cfg["G"])
    for b0, nb in ((0, 15), (1, 15), (3, 7)):
        sl = slice(b0, b0 + nb)
        rg, _, _ = O.trvae_loss_and_grads(W, cfg, xx[sl], z["batch"][sl], torch.tensor(z["sizefactor"][sl]),
                                          torch.tensor(z["eps"][sl]), alpha)
        ref += torch.tensor([rg["recon"], rg["kl"], 1.0], dtype=torch.float64)
        lo, hi = sdist.shard_range(nb, rank, WORLD)
        rl, _, _ = O.trvae_loss_and_grads(W, cfg, xx[sl][lo:hi], z["batch"][sl][lo:hi],
                                          torch.tensor(z["sizefactor"][sl][lo:hi]), torch.tensor(z["eps"][sl][lo:hi]),
                                          alpha)
        w_ = (hi - lo) / nb
        acc += w_ * torch.tensor([rl["recon"], rl["kl"], 1.0], dtype=torch.float64)
    sdist.allreduce_sum_(acc)
    res["log_err"] = float(((acc[:2] / acc[2]) - (ref[:2] / ref[2])).abs().max() / (ref[:2] / ref[2]).abs().max())
    res["log_count"] = float(acc[2])
    dist.all_gather_object(allres := [None] * WORLD, decisions)
    res["same_decisions"] = allres[0] == allres[1]
    res["stopped"] = not decisions[-1][0]
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for r in range(WORLD):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert res["cover"]
        assert res["grad_err"] < 1e-5, res           # fp32 association noise only
        assert res["grad_err_equal"] < 1e-5, res
        assert res["grad_err_mmd"] < 1e-5 and res["mmd_err"] < 1e-5, res
        assert res["same_decisions"] and res["stopped"]
        assert res["log_err"] < 1e-6 and abs(res["log_count"] - 3.0) < 1e-12, res


def test_shard_range_partitions():
    from scarches_amd.distributed import shard_range
    for n in (0, 1, 7, 128, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
