"""Model-wrapper mixins: save / load, architecture surgery (`load_query_data`) and latent mapping (`get_latent`).

Public contract of scarches/models/base/_base.py:16-387 - same method names, arguments, on-disk layout
(`model_params.pt`, `attr.pkl`, `var_names.csv`), same state-dict expansion and freeze rules - implemented on top
of the HIP engine: `get_latent` uploads the matrix once as CSR and encodes it in HBM in large row blocks instead of
512-row host slices.
"""
import inspect
import os
import pickle
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import scipy.sparse as sp
import torch

from ._utils import _validate_var_names


class BaseMixin:
    def _get_user_attributes(self):
        attrs = inspect.getmembers(self, lambda a: not inspect.isroutine(a))
        return [(n, v) for n, v in attrs
                if not (n.startswith("__") and n.endswith("__")) and not n.startswith("_abc_")]

    def _get_public_attributes(self):
        """Attributes whose name ends in '_' define the persisted configuration (attr.pkl)."""
        return {n: v for n, v in self._get_user_attributes() if n[-1] == "_"}

    def save(self, dir_path: str, overwrite: bool = False, save_anndata: bool = False, **anndata_write_kwargs):
        """Writes model_params.pt (state dict), attr.pkl (public attributes) and var_names.csv into `dir_path`."""
        if os.path.exists(dir_path) and not overwrite:
            raise ValueError(
                "{} already exists. Please provide an unexisting directory for saving.".format(dir_path))
        os.makedirs(dir_path, exist_ok=True)
        if save_anndata:
            self.adata.write(os.path.join(dir_path, "adata.h5ad"), **anndata_write_kwargs)
        np.savetxt(os.path.join(dir_path, "var_names.csv"), self.adata.var_names.astype(str).to_numpy(), fmt="%s")
        # parameters are views of one flat HBM buffer: store independent contiguous copies
        state = {k: v.detach().clone().contiguous() for k, v in self.model.state_dict().items()}
        torch.save(state, os.path.join(dir_path, "model_params.pt"))
        with open(os.path.join(dir_path, "attr.pkl"), "wb") as f:
            pickle.dump(self._get_public_attributes(), f)

    def _load_expand_params_from_dict(self, state_dict):
        """Load `state_dict` into a model that has MORE conditions: every tensor whose shape grew keeps the old block
        and takes the trailing slices of the freshly initialised tensor along the grown dimension
        (reference: _base.py:92-122)."""
        current = self.model.state_dict()
        merged = dict(state_dict)
        device = next(self.model.parameters()).device
        for key, old in state_dict.items():
            new = current[key]
            if new.size() == old.size():
                continue
            old = old.to(device)
            for ax in range(new.dim()):
                grow = new.shape[ax] - old.shape[ax]
                if grow > 0:
                    tail = new.narrow(ax, new.shape[ax] - grow, grow)
                    merged[key] = torch.cat([old, tail], dim=ax)
                    break
        for key, ten in current.items():
            merged.setdefault(key, ten)
        self.model.load_state_dict(merged)

    @classmethod
    def _load_params(cls, dir_path: str, map_location: Optional[str] = None):
        with open(os.path.join(dir_path, "attr.pkl"), "rb") as handle:
            attr_dict = pickle.load(handle)
        model_state_dict = torch.load(os.path.join(dir_path, "model_params.pt"), map_location=map_location,
                                      weights_only=True)
        var_names = np.genfromtxt(os.path.join(dir_path, "var_names.csv"), delimiter=",", dtype=str)
        return attr_dict, model_state_dict, var_names

    @classmethod
    def load(cls, dir_path: str, adata=None, map_location=None):
        """Re-instantiate a saved model on `adata` (genes are re-ordered / zero-filled to the saved var names)."""
        if adata is None:
            raise ValueError("Save path contains no saved anndata and no adata was passed.")
        attr_dict, model_state_dict, var_names = cls._load_params(dir_path, map_location)
        adata = _validate_var_names(adata, var_names)
        cls._validate_adata(adata, attr_dict)
        model = cls(adata, **cls._get_init_params_from_dict(attr_dict))
        model.model.load_state_dict(model_state_dict)
        model.model.eval()
        model.is_trained_ = attr_dict['is_trained_']
        return model


class SurgeryMixin:
    @classmethod
    def load_query_data(cls, adata, reference_model: Union[str, 'Model'], freeze: bool = True,
                        freeze_expression: bool = True, remove_dropout: bool = True, map_location=None, **kwargs):
        """Architecture surgery: extend a trained reference model by the query's new conditions and freeze everything
        but the condition-specific weights.

        freeze / freeze_expression follow the reference's parameter-NAME rules (_base.py:268-279): 'theta' stays
        trainable; with freeze_expression the 'cond_L.weight' tensors, otherwise everything named 'L0' / 'N0'.
        Note that the whole theta / cond_L tensors train (old columns get zero data gradient but still feel weight
        decay unless weight_decay=0 is passed to train()), as in the reference.
        """
        if isinstance(reference_model, str):
            attr_dict, model_state_dict, var_names = cls._load_params(reference_model, map_location)
            adata = _validate_var_names(adata, var_names)
        else:
            attr_dict = reference_model._get_public_attributes()
            model_state_dict = {k: v.detach().clone() for k, v in reference_model.model.state_dict().items()}
            adata = _validate_var_names(adata, reference_model.adata.var_names)

        init_params = deepcopy(cls._get_init_params_from_dict(attr_dict))
        conditions = init_params['conditions']
        condition_key = init_params['condition_key']
        for item in adata.obs[condition_key].unique().tolist():
            if item not in conditions:
                conditions.append(item)
        if remove_dropout:
            init_params['dr_rate'] = 0.0
        init_params.update(kwargs)

        new_model = cls(adata, **init_params)
        new_model._load_expand_params_from_dict(model_state_dict)

        if freeze:
            new_model.model.freeze = True
            for name, p in new_model.model.named_parameters():
                p.requires_grad = False
                if 'theta' in name:
                    p.requires_grad = True
                if freeze_expression:
                    if 'cond_L.weight' in name:
                        p.requires_grad = True
                else:
                    if "L0" in name or "N0" in name:
                        p.requires_grad = True
        return new_model


class CVAELatentsMixin:
    _LATENT_BLOCK = 16384     # rows encoded per launch group (the reference walks 512-row host slices)

    def _encode_labels(self, c):
        c = np.asarray(c)
        if not set(c.tolist()).issubset(self.conditions_):
            raise ValueError("Incorrect conditions")
        labels = np.zeros(c.shape[0], dtype=np.int32)
        for condition, label in self.model.condition_encoder.items():
            labels[c == condition] = label
        return labels

    def _upload(self, x, eng):
        if sp.issparse(x):
            csr = x.tocsr().astype(np.float32)
            if not csr.has_canonical_format:
                csr = csr.copy()
                csr.sum_duplicates()
        else:
            csr = sp.csr_matrix(np.asarray(x, dtype=np.float32))
        eng.bind_csr(torch.from_numpy(csr.indptr.astype(np.int64)).to(eng.device),
                     torch.from_numpy(csr.indices.astype(np.int32)).to(eng.device),
                     torch.from_numpy(csr.data.astype(np.float32)).to(eng.device))
        return csr.shape[0]

    def get_latent(self, x=None, c=None, mean: bool = False, mean_var: bool = False):
        """Latent representation of `x` (default: self.adata.X) given condition labels `c`.
        mean=True: encoder means; mean_var=True: (means, variances); otherwise a sample z."""
        if x is None and c is None:
            x = self.adata.X
            if self.conditions_ is not None:
                c = self.adata.obs[self.condition_key_]
        eng = self.model.engine()
        labels = None
        if c is not None:
            labels = torch.from_numpy(self._encode_labels(c)).to(eng.device)
        n = self._upload(x, eng)
        outs = []
        for lo in range(0, n, self._LATENT_BLOCK):
            hi = min(n, lo + self._LATENT_BLOCK)
            rows = torch.arange(lo, hi, device=eng.device, dtype=torch.int64)
            z, mu, var = eng.encode(rows, None if labels is None else labels[lo:hi].contiguous(), hi - lo, mean=False)
            if mean:
                outs.append((mu.cpu(),))
            elif mean_var:
                outs.append((mu.cpu(), var.cpu()))
            else:
                outs.append((z.cpu(),))
        result = tuple(torch.cat(parts).numpy() for parts in zip(*outs))
        return result[0] if len(result) == 1 else result

    def get_y(self, x=None, c=None):
        """Output of the first decoder layer (the 'y' space MMD can act on)."""
        if x is None and c is None:
            x = self.adata.X
            if self.conditions_ is not None:
                c = self.adata.obs[self.condition_key_]
        eng = self.model.engine()
        labels = None
        if c is not None:
            labels = torch.from_numpy(self._encode_labels(c)).to(eng.device)
        n = self._upload(x, eng)
        outs = []
        for lo in range(0, n, self._LATENT_BLOCK):
            hi = min(n, lo + self._LATENT_BLOCK)
            rows = torch.arange(lo, hi, device=eng.device, dtype=torch.int64)
            lab = None if labels is None else labels[lo:hi].contiguous()
            eng.encode(rows, lab, hi - lo, mean=False)
            outs.append(eng.decode_first_layer(lab, hi - lo).cpu())
        return torch.cat(outs).numpy()
