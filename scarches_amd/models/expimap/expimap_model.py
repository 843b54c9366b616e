"""`EXPIMAP`: the user-facing expiMap model class, API of scarches/models/expimap/expimap_model.py:18-703."""
from typing import Optional, Union

import numpy as np
import pandas as pd
import torch

from ...trainers.expimap.regularized import expiMapTrainer
from ..base._base import BaseMixin, CVAELatentsMixin, SurgeryMixin
from .expimap import expiMap


class EXPIMAP(BaseMixin, SurgeryMixin, CVAELatentsMixin):
    """Interpretable conditional VAE whose latent dimensions are gene programs (rows of `mask`, [n_terms, n_genes]).

    Constructor arguments are the reference's: adata, condition_key, conditions, hidden_layer_sizes=[256, 256],
    dr_rate=0.05, recon_loss='nb', use_l_encoder, use_bn, use_ln, mask / mask_key, decoder_last_layer, soft_mask,
    n_ext, n_ext_m, use_hsic, hsic_one_vs_all, ext_mask, soft_ext_mask.
    """

    def __init__(self, adata, condition_key: str = None, conditions: Optional[list] = None,
                 hidden_layer_sizes: list = [256, 256], dr_rate: float = 0.05, recon_loss: str = 'nb',
                 use_l_encoder: bool = False, use_bn: bool = False, use_ln: bool = True,
                 mask: Optional[Union[np.ndarray, list]] = None, mask_key: str = 'I',
                 decoder_last_layer: Optional[str] = None, soft_mask: bool = False, n_ext: int = 0, n_ext_m: int = 0,
                 use_hsic: bool = False, hsic_one_vs_all: bool = False,
                 ext_mask: Optional[Union[np.ndarray, list]] = None, soft_ext_mask: bool = False):
        self.adata = adata
        if mask is None and mask_key not in self.adata.varm:
            raise ValueError('Please provide mask.')
        self.condition_key_ = condition_key
        if conditions is None:
            self.conditions_ = adata.obs[condition_key].unique().tolist() if condition_key is not None else []
        else:
            self.conditions_ = conditions
        self.hidden_layer_sizes_ = hidden_layer_sizes
        self.dr_rate_ = dr_rate
        self.recon_loss_ = recon_loss
        self.use_bn_ = use_bn
        self.use_ln_ = use_ln
        self.input_dim_ = adata.n_vars
        self.use_l_encoder_ = use_l_encoder
        self.decoder_last_layer_ = decoder_last_layer

        if mask is None:
            mask = adata.varm[mask_key].T
        self.mask_ = mask if isinstance(mask, list) else mask.tolist()
        mask = torch.tensor(mask).float()
        self.latent_dim_ = len(self.mask_)
        self.ext_mask_ = None
        if ext_mask is not None:
            self.ext_mask_ = ext_mask if isinstance(ext_mask, list) else ext_mask.tolist()
            ext_mask = torch.tensor(ext_mask).float()
        self.n_ext_ = n_ext
        self.n_ext_m_ = n_ext_m
        self.soft_mask_ = soft_mask
        self.soft_ext_mask_ = soft_ext_mask
        self.use_hsic_ = use_hsic and n_ext > 0
        self.hsic_one_vs_all_ = hsic_one_vs_all

        self.model = expiMap(self.input_dim_, self.latent_dim_, mask, self.conditions_, self.hidden_layer_sizes_,
                             self.dr_rate_, self.recon_loss_, self.use_l_encoder_, self.use_bn_, self.use_ln_,
                             self.decoder_last_layer_, self.soft_mask_, self.n_ext_, self.n_ext_m_, self.use_hsic_,
                             self.hsic_one_vs_all_, ext_mask, self.soft_ext_mask_)
        self.is_trained_ = False
        self.trainer = None

    def train(self, n_epochs: int = 400, lr: float = 1e-3, eps: float = 0.01, alpha: Optional[float] = None,
              omega: Optional[torch.Tensor] = None, **kwargs):
        """Train with `expiMapTrainer`.  alpha: group-lasso coefficient; omega: per-term coefficients; kwargs:
        alpha_kl (default 0.35), alpha_epoch_anneal (default min(130, n_epochs)), alpha_l1, gamma_ext, beta, and the
        trVAE trainer's own."""
        if "alpha_kl" not in kwargs:
            print("The default value of alpha_kl was changed to 0.35. from 1. "
                  "This may case inconsistency with previous training results. "
                  "Set alpha_kl=1. to reproduce the previous results.")
            kwargs["alpha_kl"] = 0.35
        if "alpha_epoch_anneal" not in kwargs:
            print("alpha_epoch_anneal is used by default now. "
                  "This may case inconsistency with previous training results. "
                  "Set alpha_epoch_anneal=None to reproduce the previous results.")
            kwargs["alpha_epoch_anneal"] = min(130, n_epochs)
        self.trainer = expiMapTrainer(self.model, self.adata, alpha=alpha, omega=omega,
                                      condition_key=self.condition_key_, **kwargs)
        self.trainer.train(n_epochs, lr, eps)
        self.is_trained_ = True

    def nonzero_terms(self):
        """Boolean index of the active terms (not switched off by the group lasso)."""
        return self.model.decoder.nonzero_terms()

    def get_latent(self, x: Optional[np.ndarray] = None, c: Optional[np.ndarray] = None, only_active: bool = False,
                   mean: bool = False, mean_var: bool = False):
        """As CVAELatentsMixin.get_latent; only_active keeps the columns of active terms."""
        result = super().get_latent(x, c, mean, mean_var)
        if not only_active:
            return result
        idx = self.nonzero_terms()
        if isinstance(result, tuple):
            return tuple(r[:, idx] for r in result)
        return result[:, idx]

    # ------------------------------------------------------------------------------------------ term bookkeeping
    def _full_terms(self, terms, adata=None):
        terms = list((self.adata if adata is None else adata).uns[terms]) if isinstance(terms, str) else list(terms)
        full = self.latent_dim_ + self.n_ext_m_ + self.n_ext_
        if len(terms) != self.latent_dim_ and len(terms) != full:
            raise ValueError('The list of terms should have the same length as the mask.')
        if len(terms) == self.latent_dim_:
            terms += ['constrained_' + str(i) for i in range(self.n_ext_m_)]
            terms += ['unconstrained_' + str(i) for i in range(self.n_ext_)]
        return terms

    def update_terms(self, terms: Union[str, list] = 'terms', adata=None):
        """Append the names of extension terms to a term list (stored back into adata.uns when given by key)."""
        if isinstance(terms, str):
            adata = self.adata if adata is None else adata
            adata.uns[terms] = self._full_terms(terms, adata)
            return None
        return self._full_terms(terms)

    def term_genes(self, term: Union[str, int], terms: Union[str, list] = 'terms'):
        """DataFrame of the genes with non-zero decoder weight in `term`, sorted by |weight|."""
        names = self._full_terms(terms)
        term = names.index(term) if isinstance(term, str) else term
        L0 = self.model.decoder.L0
        n_m = self.latent_dim_ + self.n_ext_m_
        if term < self.latent_dim_:
            weights, mask_row = L0.expr_L.weight[:, term], self.mask_[term]
        elif term >= n_m:
            weights, mask_row = L0.ext_L.weight[:, term - n_m], None
        else:
            weights, mask_row = L0.ext_L_m.weight[:, term - self.latent_dim_], self.ext_mask_[term - self.latent_dim_]
        weights = weights.data.cpu().numpy()
        absw = np.abs(weights)
        order = np.argsort(absw)[::-1][:(absw > 0).sum()]
        df = pd.DataFrame({'genes': self.adata.var_names[order].tolist(), 'weights': weights[order]})
        df['in_mask'] = False if mask_row is None else np.isin(order, np.where(mask_row)[0])
        return df

    def mask_genes(self, terms: Union[str, list] = 'terms'):
        """{term: genes annotated to it in the (extension) mask}."""
        terms = list(self.adata.uns[terms]) if isinstance(terms, str) else list(terms)
        I = np.array(self.mask_)
        if self.n_ext_m_ > 0:
            I = np.concatenate((I, self.ext_mask_))
            if len(terms) == self.latent_dim_:
                terms += ['constrained_' + str(i) for i in range(self.n_ext_m_)]
            elif len(terms) == self.latent_dim_ + self.n_ext_m_ + self.n_ext_:
                terms = terms[:(self.latent_dim_ + self.n_ext_m_)]
            else:
                raise ValueError('The list of terms should have the same length as the mask.')
        I = I.astype(bool)
        return {t: self.adata.var_names[I[i]].tolist() for i, t in enumerate(terms)}

    def _all_term_weights(self):
        L0 = self.model.decoder.L0
        W = [L0.expr_L.weight.data]
        if self.n_ext_m_ > 0:
            W.append(L0.ext_L_m.weight.data)
        if self.n_ext_ > 0:
            W.append(L0.ext_L.weight.data)
        return torch.cat(W, dim=1)

    def latent_directions(self, method="sum", get_confidence=False, adata=None, key_added='directions'):
        """Sign of up-regulation per latent term ('sum' of decoder weights, or majority 'counts' with confidence)."""
        adata = self.adata if adata is None else adata
        W = self._all_term_weights()
        confidence = None
        if method == "sum":
            signs = np.sign(W.sum(0).cpu().numpy())
        elif method == "counts":
            nnz = torch.count_nonzero(W, dim=0)
            up = torch.count_nonzero(W > 0, dim=0)
            frac = (up / (nnz + (nnz == 0))).cpu().numpy()
            dead = (nnz == 0).cpu().numpy()
            confidence = np.abs(frac - 0.5) / 0.5
            confidence[dead] = 0
            signs = np.where(frac > 0.5, 1., np.where(frac < 0.5, -1., 0.))
            signs[dead] = 0
        else:
            raise ValueError("Unrecognized method for getting the latent direction.")
        adata.uns[key_added] = signs
        if get_confidence and confidence is not None:
            adata.uns[key_added + '_confindence'] = confidence

    def latent_enrich(self, groups, comparison='rest', n_sample=5000, use_directions=False,
                      directions_key='directions', select_terms=None, adata=None, exact=True, key_added='bf_scores'):
        """Bayes-factor test that the latent score of each term is larger in one group than in the comparison
        (results in adata.uns[key_added][group] = {p_h0, p_h1, bf})."""
        adata = self.adata if adata is None else adata
        if isinstance(groups, str):
            cats_col = adata.obs[groups]
            cats = cats_col.unique()
        elif isinstance(groups, dict):
            cats, cells_all = list(groups.keys()), [c for cells in groups.values() for c in cells]
            adata = adata[cells_all]
            cats_col = pd.Series(index=adata.obs_names, dtype=str)
            for g, cells in groups.items():
                cats_col[cells] = g
        else:
            raise ValueError("groups should be a string or a dict.")
        if comparison != "rest" and isinstance(comparison, str):
            comparison = [comparison]
        if comparison != "rest" and not set(comparison).issubset(cats):
            raise ValueError("comparison should be 'rest' or among the passed groups")
        directions = adata.uns[directions_key] if use_directions else None
        scores = {}
        for cat in cats:
            if cat in comparison:
                continue
            in_cat = np.asarray(cats_col == cat)
            others = ~in_cat if comparison == "rest" else np.asarray(cats_col.isin(comparison))
            ad1 = adata[in_cat][np.random.choice(in_cat.sum(), n_sample)]
            ad2 = adata[others][np.random.choice(others.sum(), n_sample)]
            z0 = self.get_latent(ad1.X, ad1.obs[self.condition_key_], mean=False, mean_var=exact)
            z1 = self.get_latent(ad2.X, ad2.obs[self.condition_key_], mean=False, mean_var=exact)
            if not exact:
                if directions is not None:
                    z0, z1 = z0 * directions, z1 * directions
                if select_terms is not None:
                    z0, z1 = z0[:, select_terms], z1[:, select_terms]
                to_reduce = z0 > z1
                zeros = (np.abs(z0).sum(0) == 0) | (np.abs(z1).sum(0) == 0)
            else:
                from scipy.special import erfc
                (m0, v0), (m1, v1) = z0, z1
                if directions is not None:
                    m0, m1 = m0 * directions, m1 * directions
                if select_terms is not None:
                    m0, m1, v0, v1 = (a[:, select_terms] for a in (m0, m1, v0, v1))
                to_reduce = 0.5 * erfc((m1 - m0) / np.sqrt(2 * (v0 + v1)))
                zeros = (np.abs(m0).sum(0) == 0) | (np.abs(m1).sum(0) == 0)
            p_h0 = np.mean(to_reduce, axis=0)
            p_h1 = 1.0 - p_h0
            bf = np.log(p_h0 + 1e-12) - np.log(p_h1 + 1e-12)
            p_h0[zeros] = p_h1[zeros] = bf[zeros] = 0
            scores[cat] = dict(p_h0=p_h0, p_h1=p_h1, bf=bf)
        adata.uns[key_added] = scores

    # ------------------------------------------------------------------------------------------ surgery
    @classmethod
    def load_query_data(cls, adata, reference_model: Union[str, 'EXPIMAP'], freeze: bool = True,
                        freeze_expression: bool = True, unfreeze_ext: bool = True, remove_dropout: bool = True,
                        new_n_ext: Optional[int] = None, new_n_ext_m: Optional[int] = None,
                        new_ext_mask: Optional[Union[np.ndarray, list]] = None, new_soft_ext_mask: bool = False,
                        **kwargs):
        """Surgery for query data: as SurgeryMixin.load_query_data, optionally adding `new_n_ext` unconstrained and
        `new_n_ext_m` constrained (masked by `new_ext_mask`) extension terms, which stay trainable together with the
        expand_{mean,var}_encoder heads when `unfreeze_ext`."""
        params = dict(adata=adata, reference_model=reference_model, freeze=freeze,
                      freeze_expression=freeze_expression, remove_dropout=remove_dropout)
        if new_n_ext is not None:
            params['n_ext'] = new_n_ext
        if new_n_ext_m is not None:
            if new_ext_mask is None:
                raise ValueError('Provide new ext_mask')
            params.update(n_ext_m=new_n_ext_m, ext_mask=new_ext_mask, soft_ext_mask=new_soft_ext_mask)
        params.update(kwargs)
        new_model = super().load_query_data(**params)
        if freeze and unfreeze_ext:
            for name, p in new_model.model.named_parameters():
                if 'ext_L.weight' in name or 'ext_L_m.weight' in name:
                    p.requires_grad = True
                if 'expand_mean_encoder' in name or 'expand_var_encoder' in name:
                    p.requires_grad = True
        return new_model

    @classmethod
    def _get_init_params_from_dict(cls, dct):
        g = dct.get
        return {
            'condition_key': dct['condition_key_'], 'conditions': dct['conditions_'],
            'hidden_layer_sizes': dct['hidden_layer_sizes_'], 'dr_rate': dct['dr_rate_'],
            'recon_loss': dct['recon_loss_'], 'use_bn': dct['use_bn_'], 'use_ln': dct['use_ln_'],
            'mask': dct['mask_'], 'decoder_last_layer': g('decoder_last_layer_', "softmax"),
            'use_l_encoder': g('use_l_encoder_', False), 'n_ext': g('n_ext_', 0), 'n_ext_m': g('n_ext_m_', 0),
            'soft_mask': g('soft_mask_', False), 'soft_ext_mask': g('soft_ext_mask_', False),
            'hsic_one_vs_all': g('hsic_one_vs_all_', False), 'use_hsic': g('use_hsic_', False),
            'ext_mask': g('ext_mask_', None),
        }

    @classmethod
    def _validate_adata(cls, adata, dct):
        if adata.n_vars != dct['input_dim_']:
            raise ValueError("Incorrect var dimension")
        if not set(adata.obs[dct['condition_key_']].unique().tolist()).issubset(dct['conditions_']):
            raise ValueError("Incorrect conditions")
