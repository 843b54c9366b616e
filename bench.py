#!/usr/bin/env python3
"""Benchmark of the north-star metric: train cells/sec (fwd + bwd + Adam) of trVAE on 2k HVGs.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic NB counts, 1M cells x 2000 HVG resident in HBM as CSR, 20 conditions,
NB loss, MMD on z, hidden [128,128], Z=10, dropout 0.05, Adam(lr 1e-3, eps 0.01, wd 0.04).  One step = one minibatch:
CSR gather -> forward -> backward -> (gradient all-reduce) -> Adam.  Cells are sharded over ranks (weak scaling:
--batch cells per GPU per step), no data-path collective except the gradient all-reduce.
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8192, help="cells per GPU per step")
    ap.add_argument("--cells", type=int, default=1_000_000, help="total cells over all GPUs")
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--conditions", type=int, default=20)
    ap.add_argument("--hidden", type=str, default="128,128")
    ap.add_argument("--latent", type=int, default=10)
    ap.add_argument("--pre-warm-steps", type=int, default=300,
                    help="untimed steps BEFORE the --warmup steps (plan recording / graph capture, allocator, and the GPU's "
                         "clock / power state: with 10 warm-up steps alone a 50-step sample reads 2 %% slower than the "
                         "steady state - 0.578 vs 0.566 ms at 8192 cells)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-families", action="store_true",
                    help="skip the short runs of the other families (expiMap, scPoli, surgery step, get_latent)")
    ap.add_argument("--no-loader-compare", action="store_true",
                    help="skip the second timed pass with the in-step loader (profiles of the headline pass only)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="gather each minibatch at the head of its own step instead of beside the previous step's tail")
    ap.add_argument("--dp-mmd", choices=["local", "exact"], default="local",
                    help="N > 1 only: MMD over each rank's slice (default: weak scaling keeps the O(B^2) MMD work per "
                         "GPU fixed) or the reference's MMD over the global minibatch (all-gather of z; O(B_global^2) "
                         "per rank - meant for the reference's batch sizes, not for 8 x 8192 cells)")
    ap.add_argument("--launch", choices=["auto", "plan", "graph", "eager"], default="auto",
                    help="recorded launch plan, HIP-graph replay, eager launches, or auto (graph up to 8192 cells on one GPU, plan above and for N > 1)")
    return ap.parse_args()


def build_trvae_params(G, C, H, Z, seed=2020):
    """Random-init weights of the reference architecture, drawn with torch's default nn.Linear init in the
    reference's module construction order (theta first, then encoder, then decoder)."""
    from collections import OrderedDict
    import torch.nn as nn
    torch.manual_seed(seed)
    P = OrderedDict()
    P["theta"] = torch.randn(G, C)
    sizes = [G] + list(H)
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        if i == 0:
            e, c = nn.Linear(a, b, bias=True), nn.Linear(C, b, bias=False)
            P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"] = e.weight.data, e.bias.data
            P["encoder.FC.L0.cond_L.weight"] = c.weight.data
        else:
            l = nn.Linear(a, b, bias=True)
            P[f"encoder.FC.L{i}.weight"], P[f"encoder.FC.L{i}.bias"] = l.weight.data, l.bias.data
    m, v = nn.Linear(H[-1], Z), nn.Linear(H[-1], Z)
    P["encoder.mean_encoder.weight"], P["encoder.mean_encoder.bias"] = m.weight.data, m.bias.data
    P["encoder.log_var_encoder.weight"], P["encoder.log_var_encoder.bias"] = v.weight.data, v.bias.data
    Hd = list(reversed(H))
    e, c = nn.Linear(Z, Hd[0], bias=False), nn.Linear(C, Hd[0], bias=False)
    P["decoder.FirstL.L0.expr_L.weight"], P["decoder.FirstL.L0.cond_L.weight"] = e.weight.data, c.weight.data
    for i in range(1, len(Hd)):
        P[f"decoder.HiddenL.L{i}.weight"] = nn.Linear(Hd[i - 1], Hd[i], bias=False).weight.data
    o = nn.Linear(Hd[-1], G)
    P["decoder.mean_decoder.0.weight"], P["decoder.mean_decoder.0.bias"] = o.weight.data, o.bias.data
    return P


def epoch_batches(cond_all, n_batches, B, gen):
    """Stratified sampling with replacement (trainers/trvae/trainer.py:164-170, WeightedRandomSampler with
    w_i = 1/(N count(cond_i))), one draw for the whole run, cut into batches, each sorted by condition."""
    counts = torch.bincount(cond_all.long())
    w = 1.0 / (cond_all.numel() * counts[cond_all.long()].double())
    ids = torch.multinomial(w, n_batches * B, replacement=True, generator=gen).view(n_batches, B)
    c = cond_all[ids]
    order = torch.argsort(c, dim=1, stable=True)
    return torch.gather(ids, 1, order).contiguous(), torch.gather(c, 1, order).contiguous()


# The REAL reference (theislab/scarches trVAE, imported from /root/reference) timed in the BUILD container by
# scripts/ref_cpu_probe.py (8 vCPU, torch 2.10 CPU, cfg2 shape G=2000 / C=20 / [128,128], fwd + bwd + Adam on an already
# collated dense minibatch, best thread count per batch size).  The reference cannot travel to the GPU box, so this is
# a stated constant with its provenance, not a measurement of this run.
REFERENCE_IN_CONTAINER = {
    "what": "real scarches trVAE (CPU, fwd+bwd+Adam, dense minibatch already collated), build container, 8 vCPU",
    "cells_per_s": {"B=128": {"threads": 1, "value": 463.0}, "B=1024": {"threads": 8, "value": 1558.0}},
    "script": "scripts/ref_cpu_probe.py",
}


def cpu_baseline(P, spec, csr_host, cond_host, budget_s=20.0):
    """The oracle (a torch-CPU restatement of the reference, oracle/cvae_oracle.py) timed on this box's host cores:
    CSR row slice + fwd + autograd bwd + Adam on minibatches of the same workload, as a small table over
    threads in {1, share/2, share} x B in {128 (the reference's default), 1024, 4096} inside one time budget, where
    `share` is the CPU share this process may use (scheduler affinity, capped by OMP_NUM_THREADS: the GPU box hands a
    one-GPU job 16 of the machine's cores).  `value` is the best cell of the table and `cores` the torch threads that
    cell used; the machine's core count, the share and the thread counts tried are stated beside it.  Checker code
    used as a *baseline only*."""
    import numpy as np
    import scipy.sparse as sp
    from oracle import cvae_oracle as O
    box_cores = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = box_cores
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    share = max(1, min(affinity, omp) if omp > 0 else affinity)
    indptr, indices, data = csr_host
    n = len(indptr) - 1
    mat = sp.csr_matrix((data, indices, indptr), shape=(n, spec["G"]))
    cfg = dict(spec)
    tried = sorted({1, max(1, share // 2), share})
    grid = [(t, B) for B in (128, 1024, 4096) for t in tried]
    per = budget_s / len(grid)
    table, best = [], None
    for threads, B in grid:
        torch.set_num_threads(threads)
        rng = np.random.default_rng(0)
        Pn = {k: v.clone() for k, v in P.items()}
        M = {k: torch.zeros_like(v) for k, v in Pn.items()}
        V = {k: torch.zeros_like(v) for k, v in Pn.items()}
        done, t, t0 = 0, 0, None
        while True:
            rows = np.sort(rng.integers(0, n, size=B))
            x = torch.tensor(mat[rows].toarray())
            eps = torch.randn(B, spec["Z"])
            _, grads, _ = O.trvae_loss_and_grads({k: v.numpy() for k, v in Pn.items()}, cfg, x, cond_host[rows],
                                                 x.sum(1), eps, 0.5, None)
            t += 1
            for k in Pn:
                O.adam_step(Pn[k], grads[k], M[k], V[k], t)
            if t0 is None:          # first step = warm-up (thread pool, allocator)
                t0 = time.time()
                continue
            done += B
            if time.time() - t0 > per:
                break
        rate = done / (time.time() - t0)
        table.append({"threads": threads, "B": B, "cells_per_s": rate, "steps": t - 1})
        if best is None or rate > best[0]:
            best = (rate, threads, B)
    return dict(value=best[0], unit="cells/s", cores=best[1], kind="port",
                box_cores=box_cores, cpu_share=share, threads_tried=tried,
                sample=f"best of {len(grid)} cells (threads x B) of ~{per:.1f} s each: B={best[2]} on {best[1]} torch "
                       f"threads of a {share}-core share of a {box_cores}-core box (CSR row slice + fwd + autograd "
                       f"bwd + Adam, dropout off)",
                table=table, reference_in_container=REFERENCE_IN_CONTAINER)


def step_roofline(G, H, Z, n_params, nnz_per_cell, B, cells_per_s_per_gpu):
    """Whole-step roofline of SURVEY.md 8(d): GEMM flops per cell (fwd + dW + dX) against the fp32 MFMA peak and the
    ideal-fusion HBM bytes per cell (logits round trip kept, Adam traffic amortised over the minibatch) against the
    HBM peak; the step's roofline is the smaller of the two rates."""
    Hd = list(reversed(H))
    F = 2.0 * (G * H[0] + sum(a * b for a, b in zip(H[:-1], H[1:])) + 2 * H[-1] * Z) \
        + 2.0 * (Z * Hd[0] + sum(a * b for a, b in zip(Hd[:-1], Hd[1:])) + Hd[-1] * G)
    flop = 3.0 * F - 2.0 * G * H[0]
    byt = 32.0 * G + 8.0 * nnz_per_cell + 28.0 * n_params / B
    r_mfma, r_hbm = 157.3e12 / flop, 8.0e12 / byt
    roof = min(r_mfma, r_hbm)
    return {"flop_per_cell": flop, "bytes_per_cell": byt, "bound": "mfma" if r_mfma <= r_hbm else "hbm",
            "mfma_cells_per_s": r_mfma, "hbm_cells_per_s": r_hbm, "roofline_cells_per_s": roof,
            "achieved_cells_per_s": cells_per_s_per_gpu, "frac": cells_per_s_per_gpu / roof,
            "achieved_tflops": cells_per_s_per_gpu * flop / 1e12, "achieved_gb_per_s": cells_per_s_per_gpu * byt / 1e9}


GEMM_ENTRY_POINTS = ("sca_gemm_f32", "sca_gemm_f32_cfg", "sca_gemm_kk_f32", "sca_gemm_group_f32")
VALU_PEAK_GINST = 1228.8    # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 vector instruction (MI355X_MICROARCH.md)


def spin(fn, ms=100.0, chunk=20):
    """Run fn back to back for ~ms milliseconds (untimed): the measurement legs below start from the GPU's steady state,
    not from what torch's sampling kernels or an idle host phase left behind (see --pre-warm-steps)."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize()


def step_plan(eng, B, rows, cond):
    """The recorded launch plan of the B-cell step (the look-ahead steady-state one when there is one); recorded here,
    on the real minibatch (rows, cond), when the timed run replayed a HIP graph."""
    def pick():
        plans = [(k, g["fwd"]) for k, g in getattr(eng, "_plans", {}).items() if k[0] == B and isinstance(g["fwd"], list)]
        plans.sort(key=lambda kg: kg[0][-1] != (True, True))
        return plans[0][1] if plans else None
    plan = pick()
    if plan is None:        # the timed run used a HIP graph / eager launches: record a plan of the same step once
        for _ in range(2):      # (the first call records, the second copies the minibatch into the plan's static buffers)
            eng.train_step(rows, cond, B, 0.5, 1e-3, use_graph="plan")
        plan = pick()
    torch.cuda.synchronize()
    return plan


def in_step_kernel_times(eng, plan, names, reps=10):
    """Duration of the named launches INSIDE the running step: the recorded plan is re-issued with a HIP event pair
    around each named launch, recorded on the stream that launch goes to (the step runs on three streams), so every
    figure includes what the kernel loses to the kernels co-running on the other streams.  Returns name -> (launches
    per step, mean microseconds per launch).  (Each event costs a few microseconds on its queue: the step measured
    this way is ~5 % longer than the timed one.)"""
    streams = {}

    def sobj(h):
        h = int(getattr(h, "value", h) or 0)
        if h not in streams:
            streams[h] = torch.cuda.default_stream(eng.device) if h == 0 else torch.cuda.ExternalStream(h, device=eng.device)
        return streams[h]
    acc = {}
    for r in range(reps + 2):
        evs = []
        for fn, args, checked in plan:
            name = getattr(fn, "__name__", "")
            if name in names:
                st = sobj(args[-1])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                rc = fn(*args)
                e1.record(st)
                evs.append((name, e0, e1))
            else:
                rc = fn(*args)
            if checked and rc:
                raise RuntimeError(f"replayed launch {name} failed: {rc}")
        torch.cuda.synchronize()
        if r >= 2:
            for name, e0, e1 in evs:
                acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    return {k: (len(v) // reps, sum(v) / len(v)) for k, v in acc.items()}


def pmc_means(B, counter):
    """kernel-name prefix -> mean counter value per dispatch, from the committed summary of the separate
    `rocprofv3 --pmc` passes of this same command (scripts/r03_pmc.sh -> profiles/r0X_pmc_summary_B<batch>.json)."""
    for f in (f"r03_pmc_summary_B{B}.json", f"r02_pmc_summary_B{B}.json", f"r01_pmc_summary_B{B}.json"):
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            try:
                with open(path) as fh:
                    pm = json.load(fh)
            except (OSError, ValueError):
                continue
            out = {k: v[counter] for k, v in pm.items() if counter in v}
            if out:
                return out, f
    return {}, None


def kernel_rates(eng, plan, B, G, nnz_per_cell, in_step, reps=20):
    """Live rates of the non-GEMM kernels the north star names, each alone on the chip (its own launch of the recorded
    step re-issued between HIP events on the launch stream) with its in-step duration beside it:
      hbm_kernels  : the CSR gather against the HBM peak (algorithmic bytes per cell: reads 8*nnz + 24, writes 4*G + 4);
      valu_kernels : the NB likelihood (`recon_rows_kernel`) and the MMD (`mmd_small_kernel`) - both bound by vector-
                     instruction issue, not by HBM: vector instructions per launch (PMC
                     SQ_INSTS_VALU of a separate committed pass; null without one) / time against the chip's issue
                     peak of one wave64 vector instruction per SIMD per 2 cycles.  The likelihood's HBM rate (reads
                     4*G logits + 8*nnz counts and columns, writes 4*G d-logits per cell) is kept as a secondary figure."""
    st = torch.cuda.current_stream(eng.device)
    hbm, valu = {}, {}

    def find(*names):
        return next(((fn, args) for fn, args, _ in plan if getattr(fn, "__name__", "") in names), None)

    def alone(fn, args, before=None):
        tot = 0.0
        for i in range(3 + reps):
            if before is not None:
                before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn(*args[:-1], st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            if i >= 3:
                tot += e0.elapsed_time(e1)
        return tot / reps * 1e3

    insts, src = pmc_means(B, "SQ_INSTS_VALU")

    def valu_entry(us, kern_prefix, step_name):
        ent = {"bound": "valu-issue", "avg_launch_us": us, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s"}
        n = next((v["mean"] for k, v in insts.items() if k.startswith(kern_prefix)), None)
        ent["valu_insts_per_launch"] = n
        ent["achieved"] = (n / us / 1e3) if n else None
        ent["frac"] = (n / us / 1e3 / VALU_PEAK_GINST) if n else None
        ent["insts_from"] = ("profiles/" + src) if (n and src) else None
        if step_name in in_step:
            ent["in_step_us"] = in_step[step_name][1]
            ent["frac_in_step"] = (n / in_step[step_name][1] / 1e3 / VALU_PEAK_GINST) if n else None
        return ent

    hit = find("sca_csr_gather_rows")
    if hit:
        us = alone(*hit)
        byt = B * (8.0 * nnz_per_cell + 24 + 4.0 * G + 4)
        hbm["csr_gather_kernel"] = {"bound": "hbm", "avg_launch_us": us, "bytes_per_launch": byt,
                                    "achieved": byt / us / 1e3, "peak": 8000.0, "unit": "GB/s",
                                    "frac": byt / us / 1e3 / 8000.0}
        if "sca_csr_gather_rows" in in_step:
            hbm["csr_gather_kernel"]["in_step_us"] = in_step["sca_csr_gather_rows"][1]
    hit = find("sca_recon_rows_tab", "sca_recon_rows")
    if hit:
        logits = eng.ws["logits"]
        keep = logits.clone()           # the kernel overwrites the logits with d-logits: restore before every launch
        us = alone(*hit, before=lambda: logits.copy_(keep))
        name = hit[0].__name__
        ent = valu_entry(us, "recon_rows_kernel", name)
        byt = B * (8.0 * G + 8.0 * nnz_per_cell)
        ent.update(hbm_bytes_per_launch=byt, hbm_gb_per_s=byt / us / 1e3, hbm_frac=byt / us / 1e3 / 8000.0,
                   note="alone on the chip; inside the step it shares the CUs with the MMD kernel")
        valu["recon_rows_kernel"] = ent
    hit = find("sca_mmd_fwd_bwd_prepared", "sca_mmd_fwd_bwd")
    if hit:
        us = alone(*hit)
        ent = valu_entry(us, "mmd_small_kernel", hit[0].__name__)
        ent.update(ordered_pairs_per_launch=float(B) * B, pairs_per_s=float(B) * B / us * 1e6,
                   note="alone on the chip; inside the step it runs beside the output GEMM and the likelihood")
        valu["mmd_small_kernel"] = ent
    return hbm, valu


# ------------------------------------------------------------------------------------------------- other families
def _lin(a, b, bias=True):
    import torch.nn as nn
    l = nn.Linear(a, b, bias=bias)
    return l.weight.data, (l.bias.data if bias else None)


def gene_program_mask(G, L, seed):
    """Binary [L, G] mask, Reactome-like: program sizes LogNormal(median 30) clipped to [13, 300], every gene in at
    least one program (density ~1.2 %)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    M = np.zeros((L, G), dtype=np.float32)
    sizes = np.clip(np.exp(rng.normal(np.log(30.0), 0.8, L)).astype(int), 13, 300)
    for l in range(L):
        M[l, rng.choice(G, sizes[l], replace=False)] = 1
    for g in np.where(M.sum(0) == 0)[0]:
        M[rng.randint(L), g] = 1
    return M


def family_runs(dev, steps, warmup, n_cells=200_000):
    """Short timed runs of the other model families and call shapes of BASELINE.json's configs, same timing protocol as
    the headline (warm-up, synchronize, K steps, synchronize), one GPU, synthetic NB counts resident in HBM as CSR:
      cfg3  expiMap, 5000 genes x 280 gene programs (hard mask, tile-skipping decoder kernels), hidden [256,256,256];
      cfg4  scPoli, 4000 genes, 200 combined conditions (40 x 5), embeddings 10 + 5, hidden [64], prototype loss on;
      cfg2' trVAE cfg2 with the class-default hidden sizes [256, 64];
      cfg5  trVAE surgery step at HLCA shape: 2000 genes, 125 conditions, only theta + the condition columns of the two
            first layers trainable (load_query_data(freeze=True, freeze_expression=True));
      get_latent : eval-mode encoder pass (CSR gather -> encoder -> mean head) over the cfg2 matrix in 8192-cell blocks."""
    from collections import OrderedDict
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    res = {}

    def timed(one, B):
        for i in range(warmup):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            one(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"ms_per_step": dt / steps * 1e3, "cells_per_s": B * steps / dt, "batch": B, "steps": steps}

    def batches(cond_all, B, gen):
        return epoch_batches(cond_all, steps + warmup + 1, B, gen)

    gen = torch.Generator(device=dev)
    gen.manual_seed(77)

    # ---- cfg3: expiMap, hard gene-program mask
    G, L, H, C, B = 5000, 280, [256, 256, 256], 4, 4096
    indptr, indices, data, cond_all, _ = make_synthetic_csr(n_cells, G, C, dev, seed=31)
    M = gene_program_mask(G, L, seed=1)
    torch.manual_seed(11)
    P = OrderedDict()
    P["theta"] = torch.randn(G, C)
    sizes = [G] + H
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        if i == 0:
            P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"] = _lin(a, b)
            P["encoder.FC.L0.cond_L.weight"] = _lin(C, b, bias=False)[0]
        else:
            P[f"encoder.FC.L{i}.weight"], P[f"encoder.FC.L{i}.bias"] = _lin(a, b)
    P["encoder.mean_encoder.weight"], P["encoder.mean_encoder.bias"] = _lin(H[-1], L)
    P["encoder.log_var_encoder.weight"], P["encoder.log_var_encoder.bias"] = _lin(H[-1], L)
    P["decoder.L0.expr_L.weight"] = _lin(L, G, bias=False)[0] * torch.tensor(M.T)
    P["decoder.L0.cond_L.weight"] = _lin(C, G, bias=False)[0]
    spec = dict(arch="expimap", G=G, C=C, H=H, Z=L, recon="nb", use_bn=False, use_ln=True, dr_rate=0.05, n_ext=0,
                n_ext_m=0, mask=torch.tensor(M.T.copy()), ext_mask=None, use_hsic=False, seed=2020)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    ids, cs = batches(cond_all, B, gen)
    r = timed(lambda i: eng.train_step(ids[i], cs[i], B, alpha_kl=0.5, lr=1e-3, eps=0.01, weight_decay=0.0), B)
    r["workload"] = (f"expiMap {G} genes x {L} gene programs (mask density {M.mean():.3f}, "
                     f"{100 * eng.masked['skipped']:.0f} % of the decoder k-steps skipped), hidden {H}, {C} conditions, NB")
    res["cfg3_expimap_train_step"] = r
    del eng, P, indptr, indices, data
    torch.cuda.empty_cache()

    # ---- cfg4: scPoli, condition embeddings + prototype loss
    G, H, Z, E, n_conds, K, B = 4000, [64], 10, [10, 5], [40, 5], 30, 4096
    Cc = n_conds[0] * n_conds[1]
    indptr, indices, data, cond_all, ctype = make_synthetic_csr(n_cells, G, Cc, dev, seed=41)
    torch.manual_seed(13)
    P = OrderedDict()
    for i, (n, e) in enumerate(zip(n_conds, E)):
        P[f"embeddings.{i}.weight"] = torch.randn(n, e)
    P["theta"] = torch.randn(G, Cc)
    Et = sum(E)
    P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"] = _lin(G, H[0])
    P["encoder.FC.L0.cond_L.weight"] = _lin(Et, H[0], bias=False)[0]
    P["encoder.mean_encoder.weight"], P["encoder.mean_encoder.bias"] = _lin(H[-1], Z)
    P["encoder.log_var_encoder.weight"], P["encoder.log_var_encoder.bias"] = _lin(H[-1], Z)
    P["decoder.FirstL.L0.expr_L.weight"] = _lin(Z, H[0], bias=False)[0]
    P["decoder.FirstL.L0.cond_L.weight"] = _lin(Et, H[0], bias=False)[0]
    P["decoder.mean_decoder.0.weight"], P["decoder.mean_decoder.0.bias"] = _lin(H[0], G)
    spec = dict(arch="scpoli", G=G, C=Cc, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, dr_rate=0.05, E=E,
                n_conds=n_conds, inject=["encoder", "decoder"], max_norm=1.0, seed=2020)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    ids, cs = batches(cond_all, B, gen)
    cs = cs.to(torch.int32)
    covs = torch.stack([cs // n_conds[1], cs % n_conds[1]], 2).to(torch.int32).contiguous()
    ct = (ctype[ids].long() % K).to(torch.int32).view(ids.shape[0], B, 1).contiguous()
    labeled = torch.ones(ids.shape[0], B, dtype=torch.uint8, device=dev)
    protos = torch.randn(K, Z, device=dev)

    def scpoli_step(i):
        eng.begin_step(alpha_kl=0.5, lr=1e-3, eps=0.01, clip=0.0)
        eng.forward_backward(ids[i], cs[i], B, covs=covs[i], proto=dict(P=protos, Pu=None, celltypes=ct[i],
                             labeled=labeled[i], eta=5.0, unlabeled_weight=0.0))
        eng.adam_step(weight_decay=0.04)
    r = timed(scpoli_step, B)
    r["workload"] = (f"scPoli {G} genes, {Cc} combined conditions ({n_conds[0]} x {n_conds[1]}), embeddings {E}, hidden "
                     f"{H}, Z={Z}, {K} labeled prototypes (eta 5), NB; eager launches (trainers/scpoli/trainer.py path)")
    res["cfg4_scpoli_train_step"] = r
    del eng, P, indptr, indices, data
    torch.cuda.empty_cache()

    # ---- cfg2 with the class-default hidden sizes [256, 64] (trvae_model.py:57)
    G, C, H, Z, B = 2000, 20, [256, 64], 10, 8192
    indptr, indices, data, cond_all, _ = make_synthetic_csr(n_cells, G, C, dev, seed=61)
    P = build_trvae_params(G, C, H, Z, seed=6)
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    ids, cs = batches(cond_all, B, gen)
    r = timed(lambda i: eng.train_step(ids[i], cs[i], B, alpha_kl=0.5, lr=1e-3, eps=0.01, weight_decay=0.04,
                                       next_rows=ids[i + 1]), B)
    r["workload"] = f"trVAE cfg2 with the class-default hidden sizes {H} (5.33 MFLOP / cell), {G} genes, {C} conditions, NB + MMD on z"
    res["cfg2_class_default_256_64_train_step"] = r
    del eng, P, indptr, indices, data
    torch.cuda.empty_cache()

    # ---- cfg5: frozen-weight surgery step + get_latent (trVAE)
    G, C, H, Z, B = 2000, 125, [128, 128], 10, 8192
    indptr, indices, data, cond_all, _ = make_synthetic_csr(n_cells, G, C, dev, seed=51)
    P = build_trvae_params(G, C, H, Z, seed=5)
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020, freeze=True)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    eng.set_trainable(["theta", "encoder.FC.L0.cond_L.weight", "decoder.FirstL.L0.cond_L.weight"])
    ids, cs = batches(cond_all, B, gen)
    r = timed(lambda i: eng.train_step(ids[i], cs[i], B, alpha_kl=0.5, lr=1e-3, eps=0.01, weight_decay=0.04,
                                       next_rows=ids[i + 1]), B)
    r["workload"] = (f"trVAE surgery step, {G} genes, {C} conditions, hidden {H}: only theta + the condition columns of "
                     f"the two first layers trainable (freeze=True, freeze_expression=True), NB + MMD on z")
    res["cfg5_trvae_frozen_surgery_step"] = r
    # get_latent: every cell once, 8192-cell blocks
    rows = torch.arange(n_cells, device=dev)
    cond32 = cond_all.to(torch.int32)
    out = torch.empty(n_cells, Z, device=dev)

    def sweep():
        for lo in range(0, n_cells, B):
            n = min(B, n_cells - lo)
            z = eng.encode(rows[lo:lo + n], cond32[lo:lo + n], n, mean=True)[0]
            out[lo:lo + n].copy_(z[:n])
    sweep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["get_latent"] = {"cells_per_s": n_cells / dt, "ms_per_block": dt / -(-n_cells // B) * 1e3, "batch": B,
                         "cells": n_cells,
                         "workload": f"trVAE get_latent (eval encoder pass, mean), {n_cells} cells x {G} genes, {C} conditions"}
    del eng, P, indptr, indices, data
    torch.cuda.empty_cache()
    return res


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box (tests/test_gpu_bench_dp.py): every rank on GPU 0, gloo instead of RCCL
    rehearsal = os.environ.get("SCA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("SCA_BENCH_HANG_S", "100")), exit=True)      # a hung collective names itself instead of timing out
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    from scarches_amd.engine import CVAEEngine
    from scarches_amd.synth import make_synthetic_csr
    from scarches_amd import distributed as sdist

    G, C, Z = a.genes, a.conditions, a.latent
    H = [int(x) for x in a.hidden.split(",")]
    B = a.batch
    n_local = a.cells // world
    indptr, indices, data, cond_all, _ = make_synthetic_csr(n_local, G, C, dev, seed=1000 + rank)
    P = build_trvae_params(G, C, H, Z)
    spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z", boundary=None,
                beta=1.0, dr_rate=0.05, seed=2020 + rank)
    eng = CVAEEngine(spec, P, dev)
    eng.bind_csr(indptr, indices, data)
    if world > 1:
        eng.set_dp(rank, world, exact_mmd=(a.dp_mmd == "exact"))
    gen = torch.Generator(device=dev)
    gen.manual_seed(2020 + rank)

    def run(Bx, steps, warmup, pre=0):
        # `pre` untimed steps in front of the `warmup` untimed steps, all drawn in ONE sampler call up front: torch's
        # sampling kernels (multinomial over 1M weights, argsort, gathers) right in front of a short warm-up leave the
        # GPU 2 - 4 % below its steady state for the first tens of milliseconds (0.578 - 0.589 vs 0.566 ms per step)
        return run_on_stream(Bx, steps, warmup + pre)

    def run_on_stream(Bx, steps, warmup):
        # (one minibatch more than steps: the loader runs one minibatch ahead - every timed step gathers one, the
        #  first timed step trains on the one the last warm-up step gathered, the last gathers one nobody trains on)
        ids, cs = epoch_batches(cond_all, steps + warmup + 1, Bx, gen)
        def one(i):
            # fwd -> bwd -> (RCCL gradient all-reduce) -> Adam, with the CSR gather of minibatch i + 1 beside the
            # gradient tail (--no-prefetch: gather of minibatch i at the head of the step instead); see --launch
            eng.train_step(ids[i], cs[i], Bx, alpha_kl=0.5, lr=1e-3, eps=0.01, clip=0.0, weight_decay=0.04,
                           world=world, use_graph={"auto": "auto", "plan": "plan", "graph": True, "eager": False}[a.launch],
                           next_rows=None if (a.no_prefetch or run.in_step_loader) else ids[i + 1])
        for i in range(warmup):
            one(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            one(i)
        run.host_issue_s = time.perf_counter() - t0     # host time to ENQUEUE the steps (the GPU runs behind it)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    run.in_step_loader = False
    dt = run(B, a.steps, a.warmup, pre=a.pre_warm_steps)
    host_issue_ms = run.host_issue_s / a.steps * 1e3
    dt_in_step = None
    from scarches_amd.engine import GRAPH_MAX_B
    eff_launch = ({"plan": "plan", "graph": "graph", "eager": "eager"}.get(a.launch)
                  or ("graph" if (B <= GRAPH_MAX_B and world == 1) else "plan"))
    if not a.no_prefetch and not a.no_loader_compare and world == 1 and eff_launch == "plan":
        # the same steps with every minibatch gathered at the head of its own step (no look-ahead): reported beside
        # the headline so that the loader's share is visible in the line itself
        run.in_step_loader = True
        dt_in_step = run(B, a.steps, a.warmup, pre=a.pre_warm_steps)
        run.in_step_loader = False
    value = B * world * a.steps / dt
    losses = eng.losses.tolist()
    extra = {}
    if world > 1:
        # the same workload under the OTHER MMD semantics, so one line shows what the reference's global-minibatch MMD
        # (all-gather of z, O(B_global^2) pair evaluations over the job) costs next to the per-rank estimator
        other = "local" if a.dp_mmd == "exact" else "exact"
        eng.set_dp(rank, world, exact_mmd=(other == "exact"))
        dt_other = run(B, a.steps, a.warmup, pre=a.pre_warm_steps)
        eng.set_dp(rank, world, exact_mmd=(a.dp_mmd == "exact"))
        extra = {f"dp_mmd_{a.dp_mmd}_ms_per_step": dt / a.steps * 1e3,
                 f"dp_mmd_{other}_ms_per_step": dt_other / a.steps * 1e3,
                 f"dp_mmd_{other}_cells_per_s": B * world * a.steps / dt_other}

    out = {
        "metric": "train cells/sec (fwd+bwd+Adam) trVAE 2k-HVG", "value": value, "unit": "cells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"trVAE synthetic {a.cells} cells x {G} HVG (CSR in HBM, "
                               f"{data.numel() / n_local:.0f} nnz/cell), {C} conditions, NB loss, MMD on z, hidden {H}, "
                               f"Z={Z}, dropout 0.05, Adam(eps 0.01, wd 0.04)",
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}" if world > 1 else "single",
                   "launch": a.launch, "launch_effective": eff_launch,
                   "loader": "one minibatch ahead" if (eff_launch == "plan" and not a.no_prefetch) else "in-step",
                   "mmd": "single device" if world == 1 else
                          ("global minibatch (all-gather of z)" if a.dp_mmd == "exact" else "per-rank slice")},
        "final_losses": {"recon": losses[0], "kl": losses[1], "mmd": losses[2]},
        "host_issue_ms_per_step": host_issue_ms,
        "pre_warm_steps": a.pre_warm_steps,
    }
    if dt_in_step is not None:
        out["in_step_loader"] = {"ms_per_step": dt_in_step / a.steps * 1e3, "value": B * a.steps / dt_in_step,
                                 "note": "same steps, CSR gather at the head of each step instead of beside the "
                                         "previous step's gradient tail"}
    if extra:
        out["extra"] = extra

    if rank == 0:
        eng._dp = None      # the replays below run on rank 0 only: no collective may be part of them
        # ---- roofline of the dominant kernel family: the fp32 MFMA GEMMs of one step, replayed under HIP events
        eng.record_gemms = []
        ids, cs = epoch_batches(cond_all, 1, B, gen)
        eng.forward_backward(ids[0], cs[0], B)
        calls = eng.record_gemms
        eng.record_gemms = None
        single = ([c for c in calls if c[0] not in ("group", "kk")]
                  + [j for c in calls if c[0] == "group" for j in c[2]])
        flops = sum(2.0 * c[8] * c[9] * c[10] for c in single)
        flops += sum(2.0 * c[1][6] * c[1][7] * c[1][8] for c in calls if c[0] == "kk")
        st = torch.cuda.current_stream(dev)
        spin(lambda: eng.replay_gemms(calls))
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            eng.replay_gemms(calls)
        e1.record(st)
        torch.cuda.synchronize()
        t_gemm = e0.elapsed_time(e1) * 1e-3 / reps
        peak = 157.3
        # HBM traffic per GEMM launch: measured offline in separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes
        # of this same command (profiles/r03_pmc_summary_B<batch>.json; FETCH_SIZE doubled as the gfx950 guide
        # prescribes for 16-byte-per-lane streaming reads); null when no pass was committed for this batch size.
        traffic = None
        # algorithmic bytes of the same launches: both operands once + every output slab once
        alg = sum(4.0 * (c[8] * c[10] + c[10] * c[9] + c[8] * c[9] * max(1, c[11])) for c in single)
        alg += sum(4.0 * (c[1][6] * c[1][8] + c[1][8] * c[1][7] + c[1][6] * c[1][7] * max(1, c[1][9]))
                   for c in calls if c[0] == "kk")
        try:
            pmc = [f for f in (f"r03_pmc_summary_B{B}.json", f"r02_pmc_summary_B{B}.json", f"r01_pmc_summary_B{B}.json")
                   if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
            with open(os.path.join(ROOT, "profiles", pmc)) as fh:
                pm = json.load(fh)
            kb = n = 0.0
            for kern, ctrs in pm.items():
                if kern.startswith("gemm_") and "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
                    d = ctrs["FETCH_SIZE"]["dispatches"]
                    kb += d * (2.0 * ctrs["FETCH_SIZE"]["mean"] + ctrs["WRITE_SIZE"]["mean"])
                    n += d
            traffic = kb / n * 1024.0 if n else None
        except (OSError, ValueError, KeyError, IndexError):
            pass
        out["roofline"] = {"bound": "mfma", "achieved": flops / t_gemm / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": flops / t_gemm / 1e12 / peak, "traffic": traffic,
                           "algorithmic_bytes": alg / len(calls),
                           "traffic_over_algorithmic": (traffic / (alg / len(calls))) if traffic else None,
                           "traffic_unit": "HBM bytes per GEMM launch (PMC, separate pass)",
                           "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32), all GEMM launches of one step",
                           "launches_per_step": len(calls), "flop_per_step": flops,
                           "avg_launch_us": t_gemm / len(calls) * 1e6,
                           "gemm_share_of_step": t_gemm / (dt / a.steps)}
        out["roofline"]["step"] = step_roofline(G, H, Z, sum(v.numel() for v in P.values()), data.numel() / n_local,
                                                B, value / world)
        if world == 1:
            plan = step_plan(eng, B, ids[0], cs[0])
            spin(lambda: eng._replay(plan))
            in_step = in_step_kernel_times(eng, plan, set(GEMM_ENTRY_POINTS) | {
                "sca_csr_gather_rows", "sca_recon_rows", "sca_recon_rows_tab", "sca_mmd_fwd_bwd_prepared",
                "sca_mmd_fwd_bwd", "sca_mid_fwd", "sca_mid_bwd"})
            t_in = sum(n * us for k, (n, us) in in_step.items() if k in GEMM_ENTRY_POINTS) * 1e-6
            n_in = sum(n for k, (n, us) in in_step.items() if k in GEMM_ENTRY_POINTS)
            if t_in > 0:
                # the same GEMM launches timed INSIDE the step (co-running kernels on the other streams included)
                out["roofline"]["frac_in_step"] = flops / t_in / 1e12 / peak
                out["roofline"]["achieved_in_step"] = flops / t_in / 1e12
                out["roofline"]["avg_launch_us_in_step"] = t_in / max(n_in, 1) * 1e6
                out["roofline"]["frac_note"] = ("frac = the step's GEMM launches replayed alone; frac_in_step = the "
                                                "same launches inside the running step (event pairs on their streams)")
            out["in_step_kernel_us"] = {k: {"launches": n, "avg_us": us} for k, (n, us) in sorted(in_step.items())}
            out["in_step_from"] = ("the timed run's own launch plan" if eff_launch == "plan" else
                                   "a launch plan of the same step recorded for this measurement (the timed run is "
                                   + eff_launch + "-launched: no per-launch events possible inside it)")
            spin(lambda: eng._replay(plan))
            out["hbm_kernels"], out["valu_kernels"] = kernel_rates(eng, plan, B, G, data.numel() / n_local, in_step)
        if not a.no_sweep and world == 1:
            sweep = {}
            for Bx in (128, 1024, 4096, 16384):
                if Bx == B:
                    continue
                dts = run(Bx, max(5, a.steps // 2), 3)
                sweep[str(Bx)] = Bx * max(5, a.steps // 2) / dts
            out["cells_per_s_by_batch"] = sweep
        if not a.no_families and world == 1:
            out["families"] = family_runs(dev, max(5, a.steps // 2), max(3, a.warmup // 2))
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(P, spec, (indptr.cpu().numpy(), indices.cpu().numpy(),
                                                         data.cpu().numpy()), cond_all.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
