"""`Adaptor` / `attach_adaptors`: the condition-specific weights of one trained condition as a stand-alone artefact.

API and on-disk layout of scarches/models/trvae/adaptors.py:9-167 (`cond_params.pt` = state-dict name -> `[rows, 1]`
column of every `cond_L.weight` / `theta` tensor, `adapt_params.pkl` = condition name + the architecture the column
belongs to), so adaptors written by the reference attach to models of this package and the other way round.

The model's parameters are views of the engine's flat HBM buffer (`EngineBackedModule`): a stored column is an
independent contiguous copy (the reference stores a view, i.e. the whole tensor's storage, in the file).  Attaching
goes through `state_dict()` / `load_state_dict()` of a freshly built model, so the engine of the returned model is
laid out for the enlarged condition set on first use.
"""
import logging
import os
import pickle
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import torch

from .trvae_model import TRVAE

logger = logging.getLogger(__name__)


def _is_cond_tensor(name: str) -> bool:
    return 'cond_L.weight' in name or 'theta' in name


class Adaptor:
    """Conditional weights of ONE condition of a trained `TRVAE` (or a directory written by `Adaptor.save`).

    Parameters: trvae_model (a `TRVAE` or a path), condition (name of the condition to take; ignored for a path).
    """
    model_type = 'trVAE'

    def __init__(self, trvae_model: Union[str, TRVAE], condition: Optional[str] = None):
        if isinstance(trvae_model, str):
            self.cond_params = torch.load(os.path.join(trvae_model, "cond_params.pt"), map_location="cpu",
                                          weights_only=True)
            with open(os.path.join(trvae_model, "adapt_params.pkl"), "rb") as handle:
                self._adapt_params = pickle.load(handle)
            self.condition = self._adapt_params['condition']
            return
        self.condition = condition
        cond_idx = trvae_model.conditions_.index(condition)       # ValueError for an unknown condition, as upstream
        self.cond_params = {name: p.detach()[:, cond_idx].unsqueeze(-1).clone()
                            for name, p in trvae_model.model.state_dict().items() if _is_cond_tensor(name)}
        self._adapt_params = {
            'condition': condition,
            'model_params': {'varnames': trvae_model.adata.var_names.tolist(),
                             'hidden_layer_sizes': trvae_model.hidden_layer_sizes_,
                             'latent_dim': trvae_model.latent_dim_,
                             'recon_loss': trvae_model.recon_loss_}}

    def _validate_params(self, varnames, init_params):
        """Architecture keys must equal the model's; differing var names only warn (adaptors.py:59-72)."""
        params = self._adapt_params['model_params'].copy()
        adaptor_varnames = np.array(params.pop('varnames'), dtype=str)
        if not np.array_equal(adaptor_varnames, np.asarray(varnames).astype(str)):
            logger.warning("var_names for adata in the model does not match var_names of adata used to train the model "
                           "of the adaptor. For valid results, the vars need to be the same and in the same order as "
                           "the adata used to train the model.")
        for k in params:
            if init_params[k] != params[k]:
                raise ValueError(f'Parameter {k} in the adaptor isn\'t equal to {k} of the model.')

    def save(self, dir_path: str, overwrite: Optional[bool] = False):
        """Write `cond_params.pt` and `adapt_params.pkl` into `dir_path` (ValueError if it exists and not overwrite)."""
        if not os.path.exists(dir_path) or overwrite:
            os.makedirs(dir_path, exist_ok=overwrite)
        else:
            raise ValueError(f"{dir_path} already exists. Please provide an unexisting directory for saving.")
        torch.save({k: v.detach().cpu().contiguous() for k, v in self.cond_params.items()},
                   os.path.join(dir_path, "cond_params.pt"))
        with open(os.path.join(dir_path, "adapt_params.pkl"), "wb") as f:
            pickle.dump(self._adapt_params, f)


def attach_adaptors(trvae_model: TRVAE, adaptors: list, only_new: bool = False):
    """Return a `TRVAE` extended by the adaptors' conditions (adaptors.py:105-167).

    Conditions the model does not know are appended in the adaptors' order; for conditions it already has the adaptor
    columns replace the model's unless `only_new`.  As upstream, that replacement is written through the state dict
    of `trvae_model` itself, i.e. the passed model takes the adaptor's columns as well.
    """
    attr_dict = trvae_model._get_public_attributes()
    init_params = deepcopy(TRVAE._get_init_params_from_dict(attr_dict))

    adpt_conditions, cond_params = [], {}
    for adaptor in adaptors:
        if isinstance(adaptor, str):
            adaptor = Adaptor(adaptor)
        adaptor._validate_params(trvae_model.adata.var_names, init_params)
        adpt_conditions.append(adaptor.condition)
        for k, p in adaptor.cond_params.items():
            p = p.detach().to("cpu", copy=True)      # adaptors may come from disk (host) or from a model in HBM
            cond_params[k] = p if k not in cond_params else torch.cat([cond_params[k], p], dim=-1)

    conditions = init_params['conditions']
    inds_exist, inds_old, inds_new = [], [], []
    for i, c in enumerate(adpt_conditions):
        if c in conditions:
            inds_exist.append(i)
            inds_old.append(conditions.index(c))
        else:
            inds_new.append(i)
    init_params['conditions'] += [adpt_conditions[i] for i in inds_new]

    new_model = TRVAE(trvae_model.adata, **init_params)
    state_dict = dict(trvae_model.model.state_dict())
    for k, ten in cond_params.items():
        cur = state_dict[k]
        ten = ten.to(cur.device, cur.dtype)
        if not only_new and inds_exist:
            cur[:, inds_old] = ten[:, inds_exist]
        if inds_new:
            state_dict[k] = torch.cat([cur, ten[:, inds_new]], dim=-1)
    new_model.model.load_state_dict(state_dict)
    return new_model
