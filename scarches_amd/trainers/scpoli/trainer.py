"""scPoli trainer on the HIP engine.

Constructor arguments, attributes (`prototypes_labeled`, `prototypes_labeled_cov`, `prototypes_unlabeled`, `logs`,
`best_epoch`, ...), hook names and the pre-training / prototype schedule are those of
scarches/trainers/scpoli/trainer.py:15-806.  What changed is where the work runs:

* minibatches are row-id vectors into a CSR matrix resident in HBM; every iteration is `engine.forward_backward`
  (embedding lookups, network, prototype losses, hand-written backward) + `engine.adam_step` with the embeddings in
  their own weight_decay=0 group (trainer.py:297-305), without host synchronisation;
* the per-epoch prototype update (trainer.py:592-632) encodes the training set in HBM in large row blocks and reduces
  per-class (count, sum z, sum z z^T) on the device in fp64 (`sca_class_moments`); under data parallelism those three
  arrays are all-reduced, so every rank holds identical prototypes;
* unlabeled prototypes are one [n_clusters, Z] device tensor with its own Adam state (the reference keeps a python
  list of [1, Z] leaf tensors and a second torch optimizer); they are initialised with sklearn KMeans on the host
  (`clustering='kmeans', n_clusters=k`).  Leiden initialisation needs scanpy and raises if it is not importable.
"""
import logging
import math
import os
import time
from collections import defaultdict

import numpy as np
import torch

from ... import _lib
from ... import distributed as sdist
from ...utils.monitor import EarlyStopping
from ..trvae.trainer import Trainer, _ParamGroups
from ._utils import make_dataset, print_progress

_BLOCK = 16384


class scPoliTrainer(Trainer):
    def __init__(self, model, adata, condition_keys=None, cell_type_keys=None, batch_size: int = 128,
                 alpha_epoch_anneal: int = None, alpha_kl: float = 1., use_early_stopping: bool = True,
                 reload_best: bool = True, early_stopping_kwargs: dict = None, labeled_indices: list = None,
                 pretraining_epochs=None, clustering: str = "leiden", clustering_res: float = 2,
                 n_clusters: int = None, unlabeled_weight: float = 0, eta: float = 1, p_prototype_loss: float = 2,
                 prototype_training: bool = True, unlabeled_prototype_training: bool = True, **kwargs):
        self.adata = adata
        self.model = model
        self.condition_keys = condition_keys
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
        self.use_stratified_sampling = kwargs.pop("use_stratified_sampling", False)
        self.weight_decay = kwargs.pop("weight_decay", 0.04)
        self.clip_value = kwargs.pop("clip_value", 0.0)
        self.n_workers = kwargs.pop("n_workers", 0)
        self.seed = kwargs.pop("seed", 2020)
        self.monitor = kwargs.pop("monitor", True)
        self.monitor_only_val = kwargs.pop("monitor_only_val", True)
        self.early_stopping = EarlyStopping(**early_stopping_kwargs)
        if p_prototype_loss != 2:
            raise NotImplementedError("p_prototype_loss != 2 is not built in the HIP path")

        torch.manual_seed(self.seed)
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("scarches_amd trains on a HIP device only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.manual_seed(self.seed)
        logging.getLogger(__name__).info("GPU available: True, GPU used: True")
        self.world = torch.distributed.get_world_size() if sdist.is_dist() else 1
        self.rank = torch.distributed.get_rank() if sdist.is_dist() else 0
        self.model.seed = self.seed + 1000003 * self.rank

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

        self.eta = eta
        self.p_prototype_loss = p_prototype_loss
        self.clustering = clustering
        self.n_clusters = n_clusters
        self.unlabeled_weight = unlabeled_weight
        self.clustering_res = clustering_res
        self.pretraining_epochs = pretraining_epochs
        self.use_early_stopping_orig = self.use_early_stopping

        self.prototypes_labeled = None
        self.prototypes_labeled_cov = None
        self.prototypes_unlabeled = None
        self.best_prototypes_labeled = None
        self.best_prototypes_labeled_cov = None
        self.best_prototypes_unlabeled = None
        self.prototype_optim = None

        self.labeled_indices = range(len(adata)) if labeled_indices is None else labeled_indices
        if self.world > 1:
            # the reference's split shuffles with the unseeded global numpy RNG (trainers/scpoli/_utils.py:140,160);
            # data-parallel ranks must build the SAME split, so they seed it alike
            # - and hand the global RNG back as they found it (the single-process path never touches it)
            rng_state = np.random.get_state()
            np.random.seed(self.seed)
        try:
            self.update_labeled_indices(self.labeled_indices)
        finally:
            if self.world > 1:
                np.random.set_state(rng_state)
        self.prototype_training = prototype_training
        self.unlabeled_prototype_training = unlabeled_prototype_training
        lab_values = self.train_data.labeled_vector.unique().tolist()
        self.any_labeled_data = 1 in lab_values
        self.any_unlabeled_data = 0 in lab_values or self.model.unknown_ct_names is not None

        if self.model.prototypes_labeled["mean"] is not None:
            self.prototypes_labeled = self.model.prototypes_labeled["mean"].to(self.device, torch.float32)
            self.prototypes_labeled_cov = self.model.prototypes_labeled["cov"].to(self.device, torch.float32)

    # ------------------------------------------------------------------------------------------ data
    def update_labeled_indices(self, labeled_indices):
        """Rebuild the train / validation datasets with a new set of labeled cells."""
        self.labeled_indices = labeled_indices
        self.train_data, self.valid_data = make_dataset(
            self.adata, train_frac=self.train_frac, condition_keys=self.condition_keys,
            cell_type_keys=self.cell_type_keys, condition_encoders=self.model.condition_encoders,
            conditions_combined_encoder=self.model.conditions_combined_encoder,
            cell_type_encoder=self.model.cell_type_encoder, labeled_indices=self.labeled_indices)

    def _batch(self, dev, ids):
        """(row ids, combined condition, covariate codes, cell types, labeled) of one minibatch, sorted by the
        combined condition (theta-gradient flushes of the likelihood kernel run per condition run)."""
        c = dev["cond"][ids]
        order = torch.argsort(c, stable=True)
        ids = ids[order].contiguous()
        return dict(rows=ids, cond=c[order].contiguous(), covs=dev["covs"][ids].contiguous(),
                    celltypes=dev["celltypes"][ids].contiguous() if "celltypes" in dev else None,
                    labeled=dev["labeled"][ids].contiguous())

    def _epoch_batches(self):
        n_train = len(self.train_data)
        if self.use_stratified_sampling:
            ids = torch.multinomial(self.sampler, self.n_samples, replacement=True, generator=self._gen)
        else:
            ids = torch.randperm(n_train, device=self.device, generator=self._gen)[:self.n_samples]
        for lo in range(0, self.n_samples, self.batch_size):
            b = ids[lo:lo + self.batch_size]
            n_global = b.numel()
            if self.world > 1:
                b = sdist.local_slice(b, self.rank, self.world)
            batch = self._batch(self._train_dev, b)
            batch["n_global"] = n_global
            yield batch

    # ------------------------------------------------------------------------------------------ loop
    def train(self, n_epochs=400, lr=1e-3, eps=0.01):
        print("Initializing dataloaders")
        self.initialize_loaders()
        begin = time.time()
        self.model.train()
        self.n_epochs = n_epochs
        self.engine = self.model.engine(self.device)
        self._fresh_optimizer()
        if self.world > 1:
            self._sync_replicas()
        # two parameter groups: embeddings without weight decay, everything else with it
        self._wd_over = {k: 0.0 for k in self.engine.names if "embedding" in k}
        self.optimizer = _ParamGroups(lr, eps, self.weight_decay)
        self._eps = eps
        tr = self._train_dev
        print("Starting training")
        for self.epoch in range(n_epochs):
            self.on_epoch_begin(lr, eps)
            self.engine.bind_csr(tr["indptr"], tr["indices"], tr["data"])
            self._zero_logs()
            for self.iter, batch_data in enumerate(self._epoch_batches()):
                self.on_iteration(batch_data)
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
        self.training_time += time.time() - begin

    def _zero_logs(self):
        self.engine.accum.zero_()
        self.engine._ensure(min(self.batch_size, len(self.train_data)))
        self.engine.ws["proto_raw"].zero_()

    def _proto_args(self, batch):
        """Prototype-loss inputs of the current phase (None during pre-training)."""
        if self.epoch < self.pretraining_epochs:
            return None
        P = self.prototypes_labeled if (self.any_labeled_data and self.prototype_training) else None
        Pu = self.prototypes_unlabeled
        if P is None and Pu is None:
            return None
        return dict(P=P, Pu=Pu, celltypes=batch["celltypes"], labeled=batch["labeled"], eta=self.eta,
                    unlabeled_weight=self.unlabeled_weight)

    def loss(self, total_batch=None, train=True):
        """cvae_loss + eta * prototype_loss (trainer.py:542-590): engine.losses = [recon, kl, eta*proto, total]."""
        b = total_batch
        return self.engine.forward_backward(b["rows"], b["cond"], b["rows"].numel(), train=train, compute_grad=train,
                                            covs=b["covs"], proto=self._proto_args(b))

    def on_iteration(self, batch_data):
        eng = self.engine
        self._set_log_weight(batch_data["rows"].numel(), batch_data["n_global"])
        eng.begin_step(alpha_kl=self.calc_alpha_coeff(), lr=self.optimizer.param_groups[0]["lr"], eps=self._eps,
                       clip=self.clip_value)
        if batch_data["rows"].numel() > 0:
            self.current_loss = self.loss(batch_data)
        else:
            eng.flat_g.zero_()
        if self.world > 1:
            sdist.allreduce_mean_(eng.flat_g, self.world,
                                  weight=batch_data["rows"].numel() / batch_data["n_global"])
        eng.adam_step(weight_decay=self.weight_decay, wd_overrides=self._wd_over)

    # ------------------------------------------------------------------------------------------ prototypes
    @torch.no_grad()
    def get_latent_train(self):
        """Sampled latent (eval mode, mean=False) of the whole (local) training set, [n_train, Z] on the device."""
        eng, tr = self.engine, self._train_dev
        eng.bind_csr(tr["indptr"], tr["indices"], tr["data"])
        n = len(self.train_data)
        out = torch.empty(n, eng.Zt, dtype=torch.float32, device=self.device)
        for lo in range(0, n, _BLOCK):
            hi = min(n, lo + _BLOCK)
            rows = torch.arange(lo, hi, device=self.device, dtype=torch.int64)
            z, _, _ = eng.encode(rows, tr["cond"][lo:hi].contiguous(), hi - lo, mean=False,
                                 covs=tr["covs"][lo:hi].contiguous())
            out[lo:hi].copy_(z)
        return out

    def update_labeled_prototypes(self, latent, labels, previous_prototypes, previous_prototypes_cov, mask=None,
                                  sel=None):
        """Per cell type present in `labels` [n, n_levels] (and listed in `mask`, if given) the mean and unbiased
        covariance of its latent rows; other prototypes keep their previous value (trainer.py:667-719)."""
        L = _lib.load()
        K, Z = self.model.n_cell_types, latent.shape[1]
        st = torch.cuda.current_stream(self.device).cuda_stream
        labels = labels.to(self.device, torch.int32).contiguous()
        mom = torch.zeros(K + K * Z + K * Z * Z, dtype=torch.float64, device=self.device)
        cnt, sm, outer = mom[:K], mom[K:K + K * Z], mom[K + K * Z:]
        _lib.check(L.sca_class_moments(latent.data_ptr(), Z, latent.shape[0], Z, labels.data_ptr(), labels.shape[1],
                                       _lib.ptr(sel), K, cnt.data_ptr(), sm.data_ptr(), outer.data_ptr(), st),
                   "class_moments")
        if self.world > 1:
            sdist.allreduce_sum_(mom)
        upd = None
        if mask is not None:
            upd = torch.zeros(K, dtype=torch.uint8, device=self.device)
            if len(mask) > 0:
                upd[torch.as_tensor(list(mask), device=self.device, dtype=torch.long)] = 1
        mean = torch.empty(K, Z, dtype=torch.float32, device=self.device)
        cov = torch.empty(K, Z, Z, dtype=torch.float32, device=self.device)
        _lib.check(L.sca_class_moments_finish(cnt.data_ptr(), sm.data_ptr(), outer.data_ptr(), K, Z,
                                              _lib.ptr(previous_prototypes), _lib.ptr(previous_prototypes_cov),
                                              _lib.ptr(upd), mean.data_ptr(), cov.data_ptr(), st), "moments_finish")
        return mean, cov

    def initialize_prototypes(self):
        latent = self.get_latent_train()
        tr = self._train_dev
        if self.any_labeled_data:
            if self.prototypes_labeled is not None:
                # surgery: prototypes of cell types the reference did not know are appended
                new = list(self.model.new_prototypes or [])
                if len(new) > 0:
                    K, Z = self.model.n_cell_types, latent.shape[1]
                    pm = torch.zeros(K, Z, device=self.device)
                    pc = torch.zeros(K, Z, Z, device=self.device)
                    k0 = self.prototypes_labeled.shape[0]
                    pm[:k0], pc[:k0] = self.prototypes_labeled, self.prototypes_labeled_cov
                    self.prototypes_labeled, self.prototypes_labeled_cov = self.update_labeled_prototypes(
                        latent, tr["celltypes"], pm, pc, new, sel=tr["labeled"])
            else:
                self.prototypes_labeled, self.prototypes_labeled_cov = self.update_labeled_prototypes(
                    latent, tr["celltypes"], None, None, None, sel=tr["labeled"])
        if self.any_unlabeled_data and self.unlabeled_prototype_training:
            lat = latent.cpu().numpy()
            if self.world > 1:
                raise NotImplementedError("unlabeled prototype clustering under data parallelism is not built")
            if self.clustering == "kmeans" and self.n_clusters is not None:
                from sklearn.cluster import KMeans
                print(f"\nInitializing unlabeled prototypes with KMeans with a given number of"
                      f"{self.n_clusters} clusters.")
                centers = KMeans(n_clusters=self.n_clusters).fit(lat).cluster_centers_
            else:
                try:
                    import scanpy as sc
                except ImportError as e:
                    raise ImportError("Leiden initialisation of unlabeled prototypes needs scanpy; pass "
                                      "clustering='kmeans', n_clusters=k or unlabeled_prototype_training=False") from e
                import pandas as pd
                ad = sc.AnnData(lat)
                sc.pp.neighbors(ad)
                sc.tl.leiden(ad, resolution=self.clustering_res)
                grp = np.asarray(ad.obs["leiden"], dtype=int)
                centers = np.asarray(pd.DataFrame(lat).groupby(grp).mean())
                self.n_clusters = centers.shape[0]
                print(f"Clustering succesful. Found {self.n_clusters} clusters.")
            self.prototypes_unlabeled = torch.tensor(centers, dtype=torch.float32, device=self.device).contiguous()

    def on_epoch_begin(self, lr, eps):
        if self.epoch == self.pretraining_epochs and self.prototype_training:
            self.initialize_prototypes()
            if self.any_unlabeled_data and self.unlabeled_prototype_training:
                # Adam state of the unlabeled prototypes (their own optimiser in the reference, trainer.py:527-532)
                pu = self.prototypes_unlabeled
                self.prototype_optim = dict(m=torch.zeros_like(pu), v=torch.zeros_like(pu), t=0, lr=lr, eps=eps,
                                            hyper=torch.zeros(8, dtype=torch.float32, device=self.device))
        if self.epoch < self.pretraining_epochs:
            self.use_early_stopping = False
        if self.use_early_stopping_orig and self.epoch >= self.pretraining_epochs:
            self.use_early_stopping = True
        if self.epoch >= self.pretraining_epochs and self.epoch - 1 == self.best_epoch:
            self.best_prototypes_labeled = self.prototypes_labeled
            self.best_prototypes_labeled_cov = self.prototypes_labeled_cov
            self.best_prototypes_unlabeled = self.prototypes_unlabeled

    def _update_unlabeled_prototypes(self, latent):
        """One Adam step on the unlabeled prototypes against the whole training latent (trainer.py:621-632)."""
        L = _lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        pu, opt = self.prototypes_unlabeled, self.prototype_optim
        n, Z, Ku = latent.shape[0], latent.shape[1], pu.shape[0]
        work = torch.empty(L.sca_proto_workspace_bytes(n, 0, Ku), dtype=torch.uint8, device=self.device)
        dP = torch.zeros_like(pu)
        _lib.check(L.sca_proto_loss(latent.data_ptr(), Z, n, Z, None, 0, None, 0, None, 0.0, pu.data_ptr(), Ku, 1.0,
                                    work.data_ptr(), None, None, 0, None, Z, dP.data_ptr(), st), "proto_update")
        opt["t"] += 1
        t = opt["t"]
        h = torch.tensor([0.0, opt["lr"], 1.0 - 0.9 ** t, math.sqrt(1.0 - 0.999 ** t), opt["eps"], 0.0, 0.0, 0.0])
        opt["hyper"].copy_(h)
        # the update runs on a fresh tensor: best_prototypes_unlabeled may alias the previous one
        pu = pu.clone()
        _lib.check(L.sca_adam_step(pu.data_ptr(), dP.data_ptr(), opt["m"].data_ptr(), opt["v"].data_ptr(),
                                   pu.numel(), opt["hyper"].data_ptr(), 0.9, 0.999, float(self.weight_decay), st),
                   "proto_adam")
        self.prototypes_unlabeled = pu

    def on_epoch_end(self):
        self.model.eval()
        # the epoch's train logs are read before the prototype pass re-uses the engine's accumulators
        self._read_logs("epoch_")
        if self.epoch >= self.pretraining_epochs and self.prototype_training:
            latent = self.get_latent_train()
            tr = self._train_dev
            if self.any_labeled_data:
                self.prototypes_labeled, self.prototypes_labeled_cov = self.update_labeled_prototypes(
                    latent, tr["celltypes"], self.prototypes_labeled, self.prototypes_labeled_cov,
                    self.model.new_prototypes, sel=tr["labeled"])
            if self.any_unlabeled_data and self.unlabeled_prototype_training:
                self._update_unlabeled_prototypes(latent)
        self.model.train()
        if self.valid_data is not None:
            self.validate()
        if self.monitor and self.rank == 0:
            print_progress(self.epoch, self.logs, self.n_epochs, self.monitor_only_val)

    def after_loop(self):
        if self.best_state_dict is not None and self.reload_best:
            self.prototypes_labeled = self.best_prototypes_labeled
            self.prototypes_labeled_cov = self.best_prototypes_labeled_cov
            self.prototypes_unlabeled = self.best_prototypes_unlabeled
        self.model.prototypes_labeled["mean"] = self.prototypes_labeled
        self.model.prototypes_labeled["cov"] = self.prototypes_labeled_cov
        self.model.prototypes_unlabeled["mean"] = self.prototypes_unlabeled

    # ------------------------------------------------------------------------------------------ logs / validation
    def _read_logs(self, prefix):
        acc = self.engine.accum.clone()
        raw = self.engine.ws["proto_raw"].to(torch.float64)
        if self.world > 1:
            sdist.allreduce_sum_(acc)
            sdist.allreduce_sum_(raw)
        acc, raw = acc.cpu().numpy(), raw.cpu().numpy()
        n = max(acc[4], 1.0)
        recon, kl, wproto, tot = acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n
        lab, unl = raw[0] / n, raw[1] / n
        self.logs[prefix + "loss"].append(tot)
        self.logs[prefix + "unweighted_loss"].append(recon + kl + lab + self.unlabeled_weight * unl)
        self.logs[prefix + "cvae_loss"].append(tot - wproto)
        if self.epoch >= self.pretraining_epochs:
            self.logs[prefix + "prototype_loss"].append(wproto)
            if self.any_unlabeled_data:
                self.logs[prefix + "unlabeled_loss"].append(unl)
            if self.any_labeled_data:
                self.logs[prefix + "labeled_loss"].append(lab)

    @torch.no_grad()
    def validate(self):
        self.model.eval()
        eng, va = self.engine, self._valid_dev
        eng.bind_csr(va["indptr"], va["indices"], va["data"])
        self._zero_logs()
        n = len(self.valid_data)
        perm = torch.randperm(n, device=self.device, generator=self._gen)
        eng.set_hyper(alpha_kl=self.calc_alpha_coeff())
        for lo in range(0, n, self._val_batch):
            b = perm[lo:lo + self._val_batch]
            if self.world > 1:
                n_global = b.numel()
                s, e = sdist.shard_range(b.numel(), self.rank, self.world)
                b = b[s:e]
                if b.numel() == 0:
                    continue
                self._set_log_weight(b.numel(), n_global)
            eng.step_ctr += 1
            self.loss(self._batch(va, b), train=False)
        self._read_logs("val_")
        self.model.train()
