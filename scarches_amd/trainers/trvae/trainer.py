"""Training loop of the conditional VAE on the HIP engine.

Constructor arguments, keyword extras, attributes (`logs`, `best_epoch`, `iters_per_epoch`, ...), hook names and the
epoch / early-stopping / LR-decay behaviour are those of scarches/trainers/trvae/trainer.py:14-325.  What changed is
the data path and the step: the AnnData matrix is uploaded once as CSR, each epoch's stratified sample is drawn on
the device in one `torch.multinomial` call (as WeightedRandomSampler does, trainer.py:164-170), cut into
condition-sorted minibatches, and every iteration is `engine.forward_backward` + `engine.adam_step` with no host
synchronisation; per-iteration losses are accumulated on the device and read once per epoch.

Data parallelism (new, the reference is single-device): when torch.distributed is initialised with world_size > 1,
each rank keeps a strided shard of the training cells, `batch_size` is the GLOBAL minibatch (batch_size // world per
rank), gradients are averaged with one all-reduce of the flat buffer per step and the epoch log sums are all-reduced
so every rank takes the same early-stopping / LR decision.  The MMD couples all cells of a minibatch: by default
(`dp_exact_mmd=True`) the ranks all-gather their latent rows and every rank evaluates the reference's MMD over the
GLOBAL minibatch (identical losses and gradients to a single-device run, CVAEEngine.set_dp); `dp_exact_mmd=False`
takes the MMD of each rank's slice instead (a per-rank estimator: no collective, O(B_local^2) work).
"""
import logging
import os
import time
from collections import defaultdict

import numpy as np
import torch

from ... import _lib
from ... import distributed as sdist
from ...utils.monitor import EarlyStopping
from ._utils import make_dataset, print_progress

LOG_KEYS = ("recon_loss", "kl_loss", "mmd_loss", "loss")


class _ParamGroups:
    """Stand-in for `trainer.optimizer` (a torch optimizer in the reference): only `param_groups[i]['lr']` is part
    of the contract (early stopping multiplies it).  Adam state lives in the engine's flat buffers."""

    def __init__(self, lr, eps, weight_decay):
        self.param_groups = [dict(lr=lr, eps=eps, weight_decay=weight_decay, betas=(0.9, 0.999))]


class Trainer:
    def __init__(self, model, adata, condition_key: str = None, cell_type_keys: str = None, batch_size: int = 128,
                 alpha_epoch_anneal: int = None, alpha_kl: float = 1., use_early_stopping: bool = True,
                 reload_best: bool = True, early_stopping_kwargs: dict = None, **kwargs):
        self.adata = adata
        self.model = model
        self.condition_key = condition_key
        self.cell_type_keys = cell_type_keys
        self.batch_size = batch_size
        self.alpha_epoch_anneal = alpha_epoch_anneal
        self.alpha_iter_anneal = kwargs.pop("alpha_iter_anneal", None)
        self.use_early_stopping = use_early_stopping
        self.reload_best = reload_best
        self.alpha_kl = alpha_kl
        early_stopping_kwargs = early_stopping_kwargs if early_stopping_kwargs else dict()

        self.n_samples = kwargs.pop("n_samples", None)
        self.train_frac = kwargs.pop("train_frac", 0.9)
        self.use_stratified_sampling = kwargs.pop("use_stratified_sampling", True)
        self.weight_decay = kwargs.pop("weight_decay", 0.04)
        self.clip_value = kwargs.pop("clip_value", 0.0)
        self.n_workers = kwargs.pop("n_workers", 0)          # accepted for compatibility; there are no host workers
        self.seed = kwargs.pop("seed", 2020)
        self.monitor = kwargs.pop("monitor", True)
        self.monitor_only_val = kwargs.pop("monitor_only_val", True)
        # how the engine issues a step: "auto" (HIP graph for small batches, recorded launch plan above), "plan",
        # True (graph) or False (eager Python launches)
        self.launch_mode = kwargs.pop("launch_mode", "auto")
        self.dp_exact_mmd = kwargs.pop("dp_exact_mmd", True)
        self.early_stopping = EarlyStopping(**early_stopping_kwargs)

        torch.manual_seed(self.seed)
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("scarches_amd trains on a HIP device only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.manual_seed(self.seed)
        logging.info("GPU available: True, GPU used: True")

        self.world = torch.distributed.get_world_size() if sdist.is_dist() else 1
        self.rank = torch.distributed.get_rank() if sdist.is_dist() else 0
        self.model.seed = self.seed + 1000003 * self.rank     # per-rank Philox streams (dropout / z noise)

        self.epoch = -1
        self.n_epochs = None
        self.iter = 0
        self.best_epoch = None
        self.best_state_dict = None
        self.current_loss = None
        self.previous_loss_was_nan = False
        self.nan_counter = 0
        self.optimizer = None
        self.training_time = 0
        self.train_data = None
        self.valid_data = None
        self.sampler = None
        self.dataloader_train = None
        self.dataloader_valid = None
        self.iters_per_epoch = None
        self.val_iters_per_epoch = None
        self.logs = defaultdict(list)

        if self.world > 1:
            # without a condition key the split shuffles with the unseeded global numpy RNG (trainers/trvae/_utils.py:
            # 128-143); data-parallel ranks must build the SAME split, so they seed it alike
            # - and hand the global RNG back as they found it (the single-process path never touches it)
            rng_state = np.random.get_state()
            np.random.seed(self.seed)
        try:
            self.train_data, self.valid_data = make_dataset(
                self.adata, train_frac=self.train_frac, condition_key=self.condition_key,
                cell_type_keys=self.cell_type_keys, condition_encoder=self.model.condition_encoder,
                cell_type_encoder=self.model.cell_type_encoder)
        finally:
            if self.world > 1:
                np.random.set_state(rng_state)

    # ------------------------------------------------------------------------------------------ data
    def initialize_loaders(self):
        """Uploads the train / validation matrices (CSR) and prepares the epoch sampler."""
        n_train = len(self.train_data)
        if self.n_samples is None or self.n_samples > n_train:
            self.n_samples = n_train
        self.iters_per_epoch = int(np.ceil(self.n_samples / self.batch_size))
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed)             # every rank draws the SAME global sample
        self._train_dev = self.train_data.to_device(self.device)
        if self.use_stratified_sampling:
            self.sampler = torch.tensor(self.train_data.stratifier_weights, device=self.device)
        if self.valid_data is not None:
            self._valid_dev = self.valid_data.to_device(self.device)
            self._val_batch = min(self.batch_size, len(self.valid_data))
            self.val_iters_per_epoch = int(np.ceil(len(self.valid_data) / self.batch_size))

    def _epoch_batches(self):
        """One epoch of minibatches: (row ids int64, conditions int32) per iteration, rows sorted by condition.
        With DP every rank draws the same global sample and keeps its contiguous slice of each minibatch."""
        n_train = len(self.train_data)
        if self.use_stratified_sampling:
            ids = torch.multinomial(self.sampler, self.n_samples, replacement=True, generator=self._gen)
        else:
            ids = torch.randperm(n_train, device=self.device, generator=self._gen)[:self.n_samples]
        cond_all = self._train_dev["cond"]
        out = []
        for lo in range(0, self.n_samples, self.batch_size):
            b = ids[lo:lo + self.batch_size]
            n_global = b.numel()
            if self.world > 1:
                b = sdist.local_slice(b, self.rank, self.world)
            c = cond_all[b]
            order = torch.argsort(c, stable=True)
            out.append((b[order].contiguous(), c[order].contiguous(), n_global))
        return out

    # ------------------------------------------------------------------------------------------ schedule
    def calc_alpha_coeff(self):
        """KL weight: linear ramp to alpha_kl over `alpha_epoch_anneal` epochs or `alpha_iter_anneal` iterations."""
        if self.alpha_epoch_anneal is not None:
            return min(self.alpha_kl * self.epoch / self.alpha_epoch_anneal, self.alpha_kl)
        if self.alpha_iter_anneal is not None:
            return min(self.alpha_kl * (self.epoch * self.iters_per_epoch + self.iter) / self.alpha_iter_anneal,
                       self.alpha_kl)
        return self.alpha_kl

    # ------------------------------------------------------------------------------------------ loop
    def train(self, n_epochs=400, lr=1e-3, eps=0.01):
        self.initialize_loaders()
        begin = time.time()
        self.model.train()
        self.n_epochs = n_epochs
        self.engine = self.model.engine(self.device)          # also syncs requires_grad -> trainable set
        self._fresh_optimizer()
        if self.world > 1:
            self.engine.set_dp(self.rank, self.world, exact_mmd=self.dp_exact_mmd)
            self._sync_replicas()
        self.optimizer = _ParamGroups(lr, eps, self.weight_decay)
        self._eps = eps
        tr = self._train_dev
        self.before_loop()
        for self.epoch in range(n_epochs):
            self.on_epoch_begin(lr, eps)
            self.engine.bind_csr(tr["indptr"], tr["indices"], tr["data"])
            self.engine.accum.zero_()
            # one minibatch of look-ahead: the engine gathers minibatch i + 1 beside the tail of step i
            batches = iter(self._epoch_batches())
            batch_data, i = next(batches, None), 0
            while batch_data is not None:
                self.iter, self._next_batch = i, next(batches, None)
                self.on_iteration(batch_data)
                batch_data, i = self._next_batch, i + 1
            self.on_epoch_end()
            if self.use_early_stopping and not self.check_early_stop():
                break
        if self.best_state_dict is not None and self.reload_best:
            print("Saving best state of network...")
            print("Best State was in Epoch", self.best_epoch)
            self.model.load_state_dict(self.best_state_dict)
        self.model.eval()
        self.after_loop()
        torch.cuda.synchronize(self.device)
        self.training_time += (time.time() - begin)

    def _fresh_optimizer(self):
        """The reference builds a new torch.optim.Adam in every train() (trainer.py:221-223): moments and the
        bias-correction step restart from zero; the engine's Philox seed follows this trainer's seed."""
        eng = self.engine
        eng.flat_m.zero_()
        eng.flat_v.zero_()
        eng.adam_t = 0
        eng.seed = int(self.model.seed)

    def _sync_replicas(self):
        """Data-parallel start state: rank 0's parameters and BatchNorm buffers are broadcast so that every replica
        starts from the same model whatever the ranks' RNG state at construction time (what DDP does in its
        constructor); the replicas then stay identical because they apply identical all-reduced gradients."""
        eng = self.engine
        sdist.broadcast_(eng.flat_p, 0)
        for k in sorted(eng.buffers):
            sdist.broadcast_(eng.buffers[k], 0)

    def before_loop(self):
        pass

    def on_epoch_begin(self, lr, eps):
        pass

    def after_loop(self):
        pass

    def loss(self, total_batch=None):
        """Forward + backward of recon + alpha * kl + mmd for one minibatch (trainers/trvae/unsupervised.py:62-71);
        the three loss values are accumulated on the device for the epoch log."""
        raise NotImplementedError

    def on_iteration(self, batch_data):
        """zero_grad / backward / clip / step of the reference (trainer.py:262-280) as ONE engine step: forward +
        hand-written backward (+ gradient all-reduce) + Adam, replayed from a recorded launch plan / HIP graph."""
        eng = self.engine
        rows, cond, n_global = batch_data
        kw = dict(alpha_kl=self.calc_alpha_coeff(), lr=self.optimizer.param_groups[0]["lr"], eps=self._eps,
                  clip=self.clip_value)
        self._set_log_weight(rows.numel(), n_global)
        if rows.numel() == 0:           # a rank whose slice of a tiny last minibatch is empty contributes zeros
            eng.begin_step(**kw)
            eng.dp_idle_step(n_global)
            eng.flat_g.zero_()
            sdist.allreduce_mean_(eng.flat_g, self.world, weight=0.0)
            eng.adam_step(weight_decay=self.weight_decay)
            return
        nxt = getattr(self, "_next_batch", None)
        self.current_loss = eng.train_step(rows, cond, rows.numel(), weight_decay=self.weight_decay,
                                           world=self.world, use_graph=self.launch_mode,
                                           next_rows=nxt[0] if nxt is not None and nxt[0].numel() == rows.numel() else None,
                                           grad_weight=rows.numel() / n_global if self.world > 1 else None,
                                           n_global=n_global if self.world > 1 else None, **kw)

    def _set_log_weight(self, n_local, n_global):
        """Data-parallel epoch logs: a rank's minibatch means enter the accumulators with weight n_local / n_global,
        so the all-reduced sums are the means over the GLOBAL minibatches also when the shards are ragged."""
        if self.world <= 1:
            return
        w = n_local / float(max(n_global, 1))
        if w != getattr(self, "_log_w", None):
            self.engine.set_hyper(log_weight=w)
            self._log_w = w

    def _read_logs(self, prefix):
        acc = self.engine.accum.clone()
        if self.world > 1:
            sdist.allreduce_sum_(acc)
        acc = acc.cpu().numpy()
        n = max(acc[4], 1.0)
        recon, kl, mmd, tot = acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n
        self.logs[prefix + "loss"].append(tot)
        self.logs[prefix + "unweighted_loss"].append(recon + kl + mmd)
        self.logs[prefix + "recon_loss"].append(recon)
        self.logs[prefix + "kl_loss"].append(kl)
        if self.model.use_mmd:
            self.logs[prefix + "mmd_loss"].append(mmd)

    def on_epoch_end(self):
        self._read_logs("epoch_")
        if self.valid_data is not None:
            self.validate()
        if self.monitor and self.rank == 0:
            print_progress(self.epoch, self.logs, self.n_epochs, self.monitor_only_val)

    @torch.no_grad()
    def validate(self):
        """Eval-mode losses over the validation cells (dropout off, z still sampled - as in the reference)."""
        self.model.eval()
        eng, va = self.engine, self._valid_dev
        eng.bind_csr(va["indptr"], va["indices"], va["data"])
        eng.accum.zero_()
        n = len(self.valid_data)
        perm = torch.randperm(n, device=self.device, generator=self._gen)
        eng.set_hyper(alpha_kl=self.calc_alpha_coeff())
        for lo in range(0, n, self._val_batch):
            b = perm[lo:lo + self._val_batch]
            n_global = b.numel()
            if self.world > 1:
                s, e = sdist.shard_range(b.numel(), self.rank, self.world)
                b = b[s:e]
                if b.numel() == 0:
                    eng.dp_idle_step(n_global)
                    continue
            self._set_log_weight(b.numel(), n_global)
            c = va["cond"][b]
            order = torch.argsort(c, stable=True)
            eng.step_ctr += 1
            eng.forward_backward(b[order].contiguous(), c[order].contiguous(), b.numel(), train=False,
                                 compute_grad=False, n_global=n_global if self.world > 1 else None)
        self._read_logs("val_")
        self.model.train()

    def check_early_stop(self):
        metric = self.early_stopping.early_stopping_metric
        if self.early_stopping.update_state(self.logs[metric][-1]):
            # NB: like the reference this stores ALIASES of the live parameters (state_dict() tensors share
            # storage), so "reloading the best state" keeps the last-epoch weights
            self.best_state_dict = self.model.state_dict()
            self.best_epoch = self.epoch
        continue_training, update_lr = self.early_stopping.step(self.logs[metric][-1])
        if update_lr:
            print(f'\nADJUSTED LR')
            for group in self.optimizer.param_groups:
                group["lr"] *= self.early_stopping.lr_factor
        return continue_training
