"""`TRVAE`: the user-facing model class, API of scarches/models/trvae/trvae_model.py:17-166."""
from typing import Optional

from ...trainers.trvae.unsupervised import trVAETrainer
from ..base._base import BaseMixin, CVAELatentsMixin, SurgeryMixin
from .trvae import trVAE


class TRVAE(BaseMixin, SurgeryMixin, CVAELatentsMixin):
    """Conditional VAE with MMD (trVAE) for reference building and query mapping.

    Parameters are those of the reference class: adata, condition_key, conditions, hidden_layer_sizes=[256, 64],
    latent_dim=10, dr_rate=0.05, use_mmd=True, mmd_on='z', mmd_boundary=None, recon_loss='nb', beta=1,
    use_bn=False, use_ln=True.  `adata` may be an AnnData or a scarches_amd.anndata_lite.CellData.
    """

    def __init__(self, adata, condition_key: str = None, conditions: Optional[list] = None,
                 hidden_layer_sizes: list = [256, 64], latent_dim: int = 10, dr_rate: float = 0.05,
                 use_mmd: bool = True, mmd_on: str = 'z', mmd_boundary: Optional[int] = None,
                 recon_loss: Optional[str] = 'nb', beta: float = 1, use_bn: bool = False, use_ln: bool = True):
        self.adata = adata
        self.condition_key_ = condition_key
        if conditions is None:
            self.conditions_ = adata.obs[condition_key].unique().tolist() if condition_key is not None else []
        else:
            self.conditions_ = conditions
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
        self.input_dim_ = adata.n_vars

        self.model = trVAE(self.input_dim_, self.conditions_, self.hidden_layer_sizes_, self.latent_dim_,
                           self.dr_rate_, self.use_mmd_, self.mmd_on_, self.mmd_boundary_, self.recon_loss_,
                           self.beta_, self.use_bn_, self.use_ln_)
        self.is_trained_ = False
        self.trainer = None

    def train(self, n_epochs: int = 400, lr: float = 1e-3, eps: float = 0.01, **kwargs):
        """Train with `trVAETrainer`; kwargs are the trainer's (batch_size, alpha_epoch_anneal, weight_decay, ...)."""
        self.trainer = trVAETrainer(self.model, self.adata, condition_key=self.condition_key_, **kwargs)
        self.trainer.train(n_epochs, lr, eps)
        self.is_trained_ = True

    @classmethod
    def _get_init_params_from_dict(cls, dct):
        keys = ['condition_key', 'conditions', 'hidden_layer_sizes', 'latent_dim', 'dr_rate', 'use_mmd', 'mmd_on',
                'mmd_boundary', 'recon_loss', 'beta', 'use_bn', 'use_ln']
        return {k: dct[k + '_'] for k in keys}

    @classmethod
    def _validate_adata(cls, adata, dct):
        if adata.n_vars != dct['input_dim_']:
            raise ValueError("Incorrect var dimension")
        if not set(adata.obs[dct['condition_key_']].unique().tolist()).issubset(dct['conditions_']):
            raise ValueError("Incorrect conditions")
