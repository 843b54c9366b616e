"""`scPoli`: the user-facing scPoli model class, API of scarches/models/scpoli/scpoli_model.py:21-971."""
from collections import defaultdict
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch

from ...dataset import MultiConditionAnnotatedDataset
from ...trainers.scpoli import scPoliTrainer
from ..base._base import BaseMixin
from ..base._utils import _validate_var_names
from .scpoli import scpoli

_BLOCK = 16384


class scPoli(BaseMixin):
    """Conditional VAE with learnt condition embeddings and cell-type prototypes (population-level integration and
    label transfer).  Constructor arguments are the reference's (scpoli_model.py:75-102)."""

    def __init__(self, adata, share_metadata: Optional[bool] = True, obs_metadata: Optional[pd.DataFrame] = None,
                 condition_keys: Optional[Union[list, str]] = None, conditions: Optional[list] = None,
                 conditions_combined: Optional[list] = None, inject_condition: Optional[list] = ["encoder", "decoder"],
                 cell_type_keys: Optional[Union[str, list]] = None, cell_types: Optional[dict] = None,
                 unknown_ct_names: Optional[list] = None, labeled_indices: Optional[list] = None,
                 prototypes_labeled: Optional[dict] = None, prototypes_unlabeled: Optional[dict] = None,
                 hidden_layer_sizes: list = None, latent_dim: int = 10, embedding_dims: Union[list, int] = 10,
                 embedding_max_norm: float = 1.0, dr_rate: float = 0.05, use_mmd: bool = False, mmd_on: str = "z",
                 mmd_boundary: Optional[int] = None, recon_loss: Optional[str] = "nb", beta: float = 1,
                 use_bn: bool = False, use_ln: bool = True):
        self.adata = adata
        self.share_metadata_ = share_metadata
        self.condition_keys_ = [condition_keys] if isinstance(condition_keys, str) else condition_keys
        self.cell_type_keys_ = [cell_type_keys] if isinstance(cell_type_keys, str) else cell_type_keys
        if unknown_ct_names is not None and type(unknown_ct_names) is not list:
            raise TypeError(f"Parameter 'unknown_ct_names' has to be list not {type(unknown_ct_names)}")
        self.unknown_ct_names_ = unknown_ct_names
        self.labeled_indices_ = range(len(adata)) if labeled_indices is None else labeled_indices

        if conditions is None:
            self.conditions_ = {}
            if condition_keys is not None:
                for cond in self.condition_keys_:
                    self.conditions_[cond] = adata.obs[cond].unique().tolist()
        else:
            self.conditions_ = conditions

        if conditions_combined is None:
            self._set_combined(self.adata, self.condition_keys_)
            self.conditions_combined_ = self.adata.obs['conditions_combined'].unique().tolist()
        else:
            self.conditions_combined_ = conditions_combined

        if obs_metadata is not None:
            self.obs_metadata_ = obs_metadata
        elif self.share_metadata_ is True:
            self.obs_metadata_ = adata.obs.groupby('conditions_combined').first()
        else:
            self.obs_metadata_ = []
        if self.share_metadata_:
            self.obs_metadata_ = adata.obs.groupby(self.condition_keys_).first()

        # cell types: {name: [hierarchy keys it appears in]} over the labeled cells
        if cell_types is None:
            self.cell_types_ = dict()
            if cell_type_keys is not None:
                lab = np.asarray(list(self.labeled_indices_))
                for key in self.cell_type_keys_:
                    for ct in adata.obs[key].iloc[lab].unique().tolist():
                        self.cell_types_.setdefault(ct, []).append(key)
        else:
            self.cell_types_ = cell_types
        if self.unknown_ct_names_ is not None:
            for unknown_ct in self.unknown_ct_names_:
                self.cell_types_.pop(unknown_ct, None)

        if hidden_layer_sizes is None:
            self.hidden_layer_sizes_ = [int(np.ceil(np.sqrt(adata.shape[1])))]
        else:
            self.hidden_layer_sizes_ = hidden_layer_sizes
        self.latent_dim_ = latent_dim
        self.dr_rate_ = dr_rate
        self.use_mmd_ = use_mmd
        self.mmd_on_ = mmd_on
        self.mmd_boundary_ = mmd_boundary
        self.recon_loss_ = recon_loss
        self.beta_ = beta
        self.use_bn_ = use_bn
        self.use_ln_ = use_ln
        self.inject_condition_ = inject_condition
        if isinstance(embedding_dims, int):
            self.embedding_dims_ = [embedding_dims] * len(self.condition_keys_)
        else:
            assert len(embedding_dims) == len(self.condition_keys_), \
                "Embedding dimensions passed do not match condition keys"
            self.embedding_dims_ = embedding_dims
        self.embedding_max_norm_ = embedding_max_norm
        self.input_dim_ = adata.n_vars
        self.prototypes_labeled_ = {"mean": None, "cov": None} if prototypes_labeled is None else prototypes_labeled
        self.prototypes_unlabeled_ = {"mean": None} if prototypes_unlabeled is None else prototypes_unlabeled

        self.model_cell_types = list(self.cell_types_.keys())
        self.is_trained_ = False
        self.trainer = None
        self.model = scpoli(input_dim=self.input_dim_, conditions=self.conditions_,
                            conditions_combined=self.conditions_combined_, cell_types=self.model_cell_types,
                            inject_condition=self.inject_condition_, embedding_dims=self.embedding_dims_,
                            embedding_max_norm=self.embedding_max_norm_, unknown_ct_names=self.unknown_ct_names_,
                            prototypes_labeled=self.prototypes_labeled_,
                            prototypes_unlabeled=self.prototypes_unlabeled_,
                            hidden_layer_sizes=self.hidden_layer_sizes_, latent_dim=self.latent_dim_,
                            dr_rate=self.dr_rate_, recon_loss=self.recon_loss_, beta=self.beta_, use_bn=self.use_bn_,
                            use_ln=self.use_ln_)

    @staticmethod
    def _set_combined(adata, condition_keys):
        if len(condition_keys) > 1:
            adata.obs['conditions_combined'] = adata.obs[condition_keys].apply(lambda r: '_'.join(r), axis=1)
        else:
            adata.obs['conditions_combined'] = adata.obs[condition_keys[0]].values

    def train(self, n_epochs: int = 100, pretraining_epochs=None, eta: float = 1, lr: float = 1e-3, eps: float = 0.01,
              alpha_epoch_anneal=1e2, reload_best: bool = False, prototype_training: Optional[bool] = True,
              unlabeled_prototype_training: Optional[bool] = True, **kwargs):
        """Train with `scPoliTrainer`: `pretraining_epochs` (default 90 % of n_epochs) without, the rest with the
        prototype loss weighted by `eta`."""
        self.prototype_training_ = prototype_training
        self.unlabeled_prototype_training_ = unlabeled_prototype_training
        if self.cell_type_keys_ is None:
            pretraining_epochs = n_epochs
            self.prototype_training_ = False
            print("The model is being trained without using prototypes.")
        elif pretraining_epochs is None:
            pretraining_epochs = int(np.floor(n_epochs * 0.9))
        self.trainer = scPoliTrainer(self.model, self.adata, labeled_indices=self.labeled_indices_,
                                     pretraining_epochs=pretraining_epochs, condition_keys=self.condition_keys_,
                                     cell_type_keys=self.cell_type_keys_, reload_best=reload_best,
                                     prototype_training=self.prototype_training_,
                                     unlabeled_prototype_training=self.unlabeled_prototype_training_, eta=eta,
                                     alpha_epoch_anneal=alpha_epoch_anneal, **kwargs)
        self.trainer.train(n_epochs, lr, eps)
        self.is_trained_ = True
        self.prototypes_labeled_ = self.model.prototypes_labeled
        self.prototypes_unlabeled_ = self.model.prototypes_unlabeled

    # ------------------------------------------------------------------------------------------ mapping
    def _encode_covariates(self, obs):
        cols = []
        for cond in self.condition_keys_:
            q = np.asarray(obs[cond].values)
            if not set(q.tolist()).issubset(self.conditions_[cond]):
                raise ValueError("Incorrect conditions")
            lab = np.zeros(q.shape[0], dtype=np.int32)
            for name, code in self.model.condition_encoders[cond].items():
                lab[q == name] = code
            cols.append(lab)
        return np.stack(cols, 1)

    def _latent_blocks(self, x, covs, mean):
        eng = self.model.engine()
        csr = x.tocsr().astype(np.float32) if sp.issparse(x) else sp.csr_matrix(np.asarray(x, dtype=np.float32))
        if not csr.has_canonical_format:
            csr = csr.copy()
            csr.sum_duplicates()
        eng.bind_csr(torch.from_numpy(csr.indptr.astype(np.int64)).to(eng.device),
                     torch.from_numpy(csr.indices.astype(np.int32)).to(eng.device),
                     torch.from_numpy(csr.data.astype(np.float32)).to(eng.device))
        covs = torch.from_numpy(covs).to(eng.device)
        n = csr.shape[0]
        out = torch.empty(n, eng.Zt, device=eng.device)
        for lo in range(0, n, _BLOCK):
            hi = min(n, lo + _BLOCK)
            rows = torch.arange(lo, hi, device=eng.device, dtype=torch.int64)
            z, mu, _ = eng.encode(rows, None, hi - lo, mean=False, covs=covs[lo:hi].contiguous())
            out[lo:hi].copy_(mu if mean else z)
        return out

    def get_latent(self, adata, mean: bool = False):
        """Latent representation (sample, or mean with mean=True) of `adata` as a numpy array [n_obs, latent_dim]."""
        return self._latent_blocks(adata.X, self._encode_covariates(adata.obs), mean).cpu().numpy()

    def get_conditional_embeddings(self):
        """Learnt condition embeddings: a CellData per condition key (obs index = condition names); a single object
        when the model has one condition key."""
        from ...anndata_lite import CellData
        out = {}
        for i, cond in enumerate(self.conditions_.keys()):
            W = self.model.embeddings[i].weight.detach().cpu().numpy()
            out[cond] = CellData(W, pd.DataFrame({cond: self.conditions_[cond]}))
        return out[self.condition_keys_[0]] if len(self.condition_keys_) == 1 else out

    def classify(self, adata, prototype=False, p=2, get_prob=False, log_distance=True, scale_uncertainties=False):
        """Label transfer with the labeled prototypes: per cell-type key {'preds', 'uncert', 'weighted_distances'}.
        `adata` is an AnnData-like object, or a [n, latent_dim] array of latent points when prototype=True."""
        assert self.prototypes_labeled_['mean'] is not None, "Model was trained without prototypes"
        self.model.eval()
        eng = self.model.engine()
        if prototype:
            latent = torch.as_tensor(np.asarray(adata), dtype=torch.float32, device=eng.device)
        else:
            latent = self._latent_blocks(adata.X, self._encode_covariates(adata.obs), False)
        inv = {v: k for k, v in self.model.cell_type_encoder.items()}
        results = dict()
        for key in self.cell_type_keys_:
            idx = torch.tensor([i for i, ct in enumerate(self.cell_types_.keys()) if key in self.cell_types_[ct]],
                               device=eng.device)
            pred, uncert, dist = self.model.classify(latent, prototype=True, classes_list=idx, p=p,
                                                     get_prob=get_prob, log_distance=log_distance)
            uncert = uncert.cpu().numpy()
            if scale_uncertainties is True:
                from sklearn.preprocessing import MinMaxScaler, RobustScaler
                uncert = RobustScaler().fit_transform(uncert.reshape(-1, 1))
                uncert = MinMaxScaler(feature_range=(0, 1)).fit_transform(uncert).reshape(-1)
            results[key] = {"preds": np.array([inv[int(v)] for v in pred.cpu().numpy()]), "uncert": uncert,
                            "weighted_distances": dist.cpu().numpy()}
        return results

    def get_prototypes_info(self, prototype_set="labeled"):
        """Prototype means as an AnnData-like table (X = [n_prototypes, latent_dim]) with `batch`, and per cell-type
        key the prototype's name, predicted label and uncertainty (scpoli_model.py:605-664).  As upstream the name
        column holds the prototype's index (its branch for true names tests `prototype_set == "l"`, never taken), and
        an unknown `prototype_set` prints a message and returns None."""
        import pandas as pd
        from ...anndata_lite import CellData
        if prototype_set == "labeled":
            prototypes, batch_name = self.prototypes_labeled_["mean"], "prototype-Set Labeled"
        elif prototype_set == "unlabeled":
            prototypes, batch_name = self.prototypes_unlabeled_["mean"], "prototype-Set Unlabeled"
        else:
            print(f"Parameter 'prototype_set' has either to be 'labeled' for labeled prototype set or 'unlabeled' "
                  f"for the unlabeled prototype set. But given value was {prototype_set}")
            return None
        prototypes = prototypes.detach().cpu().numpy()
        obs = pd.DataFrame({"batch": np.array(prototypes.shape[0] * [batch_name])})
        results = self.classify(prototypes, prototype=True)
        for key in self.cell_type_keys_:
            obs[key] = np.array([f"{i}" for i in range(prototypes.shape[0])])
            obs[key + "_pred"] = results[key]["preds"]
            obs[key + "_uncert"] = results[key]["uncert"]
        return CellData(prototypes, obs, var_names=[f"{i}" for i in range(prototypes.shape[1])])

    def add_new_cell_type(self, cell_type_name, obs_key, prototypes, x=None, c=None):
        """Promote a set of unlabeled prototypes to a new labeled cell type `cell_type_name` at hierarchy `obs_key`."""
        if x is None and c is None:
            x, obs = self.adata.X, self.adata.obs
        else:
            obs = pd.DataFrame({k: np.asarray(v) for k, v in c.items()})
        latent = self._latent_blocks(x, self._encode_covariates(obs), False)
        idx = torch.tensor([i for i, ct in enumerate(self.cell_types_.keys()) if obs_key in self.cell_types_[ct]],
                           device=latent.device)
        self.model.add_new_cell_type(latent, cell_type_name, prototypes, idx)
        self.prototypes_labeled_ = self.model.prototypes_labeled
        self.prototypes_unlabeled_ = self.model.prototypes_unlabeled
        self.cell_types_[cell_type_name] = [obs_key]

    def get_recon_loss(self, adata, batch_size: int = 128, condition_encoders: Optional[dict] = None,
                       conditions_combined_encoder: Optional[dict] = None):
        """Mean reconstruction loss of `adata` under (optionally substituted) condition encoders - used by
        shot_surgery to pick the closest reference condition."""
        ds = MultiConditionAnnotatedDataset(
            adata, self.condition_keys_,
            self.model.condition_encoders if condition_encoders is None else condition_encoders,
            self.model.conditions_combined_encoder if conditions_combined_encoder is None
            else conditions_combined_encoder)
        eng = self.model.engine()
        dev = ds.to_device(eng.device)
        eng.bind_csr(dev["indptr"], dev["indices"], dev["data"])
        # like the reference (scpoli_model.py:919-947) the model's current mode decides whether dropout is active;
        # shot_surgery switches to eval() before it calls this
        total, n = 0.0, len(ds)
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            rows = torch.arange(lo, hi, device=eng.device, dtype=torch.int64)
            eng.forward_backward(rows, dev["cond"][lo:hi].contiguous(), hi - lo, train=self.model.training,
                                 compute_grad=False, accumulate_logs=False, covs=dev["covs"][lo:hi].contiguous())
            total += float(eng.losses[0]) * (hi - lo)
        return total / n

    # ------------------------------------------------------------------------------------------ persistence / surgery
    @classmethod
    def _validate_adata(cls, adata, dct):
        if adata.n_vars != dct["input_dim_"]:
            raise ValueError("Incorrect var dimension")

    @classmethod
    def _get_init_params_from_dict(cls, dct):
        keys = ["share_metadata", "obs_metadata", "condition_keys", "conditions", "conditions_combined",
                "cell_type_keys", "cell_types", "labeled_indices", "prototypes_labeled", "prototypes_unlabeled",
                "hidden_layer_sizes", "latent_dim", "dr_rate", "use_mmd", "mmd_on", "mmd_boundary", "recon_loss",
                "beta", "use_bn", "use_ln", "embedding_dims", "embedding_max_norm", "inject_condition"]
        return {k: dct[k + "_"] for k in keys}

    @classmethod
    def load_query_data(cls, adata, reference_model: Union[str, "scPoli"], labeled_indices: Optional[list] = None,
                        unknown_ct_names: Optional[list] = None, freeze: bool = True, freeze_expression: bool = True,
                        remove_dropout: bool = True, return_new_conditions: bool = False, map_location=None):
        """Surgery for query data: new conditions get new embedding rows and theta columns; with `freeze` everything
        but the embeddings and theta is frozen (scpoli_model.py:821-834 - `freeze_expression` never re-enables the
        condition weights there, and neither does it here)."""
        if isinstance(reference_model, str):
            attr_dict, model_state_dict, var_names = cls._load_params(reference_model, map_location)
            adata = _validate_var_names(adata, var_names)
        else:
            attr_dict = deepcopy(reference_model._get_public_attributes())
            model_state_dict = {k: v.detach().clone() for k, v in reference_model.model.state_dict().items()}
        init_params = cls._get_init_params_from_dict(attr_dict)

        conditions = init_params["conditions"]
        n_reference_conditions = len(conditions)
        condition_keys = init_params["condition_keys"]
        new_conditions = defaultdict(list)
        for cond in condition_keys:
            for item in adata.obs[cond].unique():
                if item not in conditions[cond]:
                    new_conditions[cond].append(item)
                    conditions[cond].append(item)
        conditions_combined = init_params["conditions_combined"]
        cls._set_combined(adata, condition_keys)
        for item in adata.obs['conditions_combined'].unique().tolist():
            if item not in conditions_combined:
                conditions_combined.append(item)
        new_meta = adata.obs.groupby('conditions_combined').first()
        old_meta = attr_dict["obs_metadata_"]
        init_params["obs_metadata"] = pd.concat([old_meta, new_meta]) if isinstance(old_meta, pd.DataFrame) \
            else new_meta

        cell_types = init_params["cell_types"]
        cell_type_keys = init_params["cell_type_keys"]
        if cell_type_keys is not None:
            lab = np.arange(len(adata)) if labeled_indices is None else np.asarray(list(labeled_indices), dtype=int)
            found = dict()
            for key in cell_type_keys:
                for ct in adata.obs[key].iloc[lab].unique().tolist():
                    found.setdefault(ct, []).append(key)
            for unknown_ct in (unknown_ct_names or []):
                found.pop(unknown_ct, None)
            for ct, keys in found.items():
                cell_types.setdefault(ct, keys)
        if remove_dropout:
            init_params["dr_rate"] = 0.0
        init_params["labeled_indices"] = labeled_indices
        init_params["unknown_ct_names"] = unknown_ct_names
        new_model = cls(adata, **init_params)
        new_model.model.n_reference_conditions = n_reference_conditions
        new_model._load_expand_params_from_dict(model_state_dict)

        if freeze:
            new_model.model.freeze = True
            for name, p in new_model.model.named_parameters():
                p.requires_grad = "embedding" in name or "theta" in name
        return (new_model, new_conditions) if return_new_conditions else new_model

    @classmethod
    def shot_surgery(cls, adata, reference_model: Union[str, "scPoli"], labeled_indices: Optional[list] = None,
                     unknown_ct_names: Optional[list] = None, train_epochs: int = 0, batch_size: int = 128,
                     subsample: float = 1., force_cuda: bool = True, **kwargs):
        """Zero/few-shot surgery: every new condition copies the embedding (and theta column) of the reference
        condition under which its cells reconstruct best; optional fine-tuning for `train_epochs`."""
        assert 0. < subsample <= 1.
        model, new_conditions = cls.load_query_data(adata, reference_model, labeled_indices, unknown_ct_names,
                                                    freeze=train_epochs > 0, return_new_conditions=True)
        assert len(model.condition_keys_) == 1
        model.model.eval()
        cond_key = model.condition_keys_[0]
        new_list = new_conditions[cond_key]
        mapping = {}
        for new_cond in new_list:
            print(f"Processing {new_cond}.")
            sub = adata[np.asarray(adata.obs[cond_key] == new_cond)]
            if subsample < 1.:
                sub = sub[np.random.choice(len(sub), int(subsample * len(sub)), replace=False)]
            best = None
            for old_cond in model.conditions_[cond_key]:
                if old_cond in new_list:
                    continue
                enc = deepcopy(model.model.condition_encoders)
                comb = deepcopy(model.model.conditions_combined_encoder)
                enc[cond_key][new_cond] = enc[cond_key][old_cond]
                comb[new_cond] = comb[old_cond]
                rl = model.get_recon_loss(sub, batch_size, enc, comb)
                print(f"  Using {old_cond}, recon loss {rl}.")
                if best is None or rl < best:
                    best, mapping[new_cond] = rl, old_cond
        enc = model.model.condition_encoders[cond_key]
        comb = model.model.conditions_combined_encoder
        emb = model.model.embeddings[0].weight.data
        theta = None if model.model.theta is None else model.model.theta.data
        for new_cond, old_cond in mapping.items():
            emb[enc[new_cond]] = emb[enc[old_cond]]
            if theta is not None:
                theta[:, comb[new_cond]] = theta[:, comb[old_cond]]
        if train_epochs > 0:
            model.train(n_epochs=train_epochs, **kwargs)
        return model, mapping

    def _load_expand_params_from_dict(self, state_dict):
        """Embedding tables grow along dim 0 (new conditions = new rows), every other grown tensor along its last dim
        (new theta columns); the new slices keep their fresh initialisation (scpoli_model.py:949-971)."""
        current = self.model.state_dict()
        merged = dict(state_dict)
        device = next(self.model.parameters()).device
        for key, old in state_dict.items():
            new = current[key]
            if new.size() == old.size():
                continue
            old = old.to(device)
            if "embedding" in key:
                grow = new.size(0) - old.size(0)
                merged[key] = torch.cat([old, new[-grow:, ...]], dim=0)
            else:
                grow = new.size(-1) - old.size(-1)
                merged[key] = torch.cat([old, new[..., -grow:]], dim=-1)
        self.model.load_state_dict(merged)
