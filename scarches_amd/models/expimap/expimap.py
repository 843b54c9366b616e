"""`expiMap`: the network object behind `EXPIMAP.model`, API of scarches/models/expimap/expimap.py:15-222.

Same constructor, attributes (`mask`, `ext_mask`, `n_ext_decoder`, `use_hsic`, ...), parameter / buffer names and
`forward(x, batch, sizefactor, labeled) -> (recon_loss, kl_div, hsic_loss)`; the arithmetic runs in the HIP engine
(arch="expimap": fused [mean|expand] heads, ONE packed masked decoder GEMM, condition offsets folded into the
likelihood kernel).
"""
from typing import Optional

import torch

from ..base._engine_module import EngineBackedModule
from .modules import ExtEncoder, MaskedLinearDecoder


class expiMap(EngineBackedModule):
    def __init__(self,
                 input_dim: int,
                 latent_dim: int,
                 mask: torch.Tensor,
                 conditions: list,
                 hidden_layer_sizes: list = [256, 256],
                 dr_rate: float = 0.05,
                 recon_loss: str = 'nb',
                 use_l_encoder: bool = False,
                 use_bn: bool = False,
                 use_ln: bool = True,
                 decoder_last_layer: Optional[str] = None,
                 soft_mask: bool = False,
                 n_ext: int = 0,
                 n_ext_m: int = 0,
                 use_hsic: bool = False,
                 hsic_one_vs_all: bool = False,
                 ext_mask: Optional[torch.Tensor] = None,
                 soft_ext_mask: bool = False):
        super().__init__()
        assert isinstance(hidden_layer_sizes, list)
        assert isinstance(latent_dim, int)
        assert isinstance(conditions, list)
        assert recon_loss in ["mse", "nb"], "'recon_loss' must be 'mse' or 'nb'"

        print("\nINITIALIZING NEW NETWORK..............")
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.n_conditions = len(conditions)
        self.conditions = conditions
        self.condition_encoder = {k: v for k, v in zip(conditions, range(len(conditions)))}
        self.cell_type_encoder = None
        self.recon_loss = recon_loss
        self.freeze = False
        self.use_bn = use_bn
        self.use_ln = use_ln
        self.use_mmd = False

        self.n_ext_encoder = n_ext + n_ext_m
        self.n_ext_decoder = n_ext
        self.n_ext_m_decoder = n_ext_m
        self.use_hsic = use_hsic and self.n_ext_decoder > 0
        self.hsic_one_vs_all = hsic_one_vs_all
        self.hsic_beta = 1.0                     # set by the trainer (expiMapTrainer.beta)
        self.soft_mask = soft_mask and mask is not None
        self.soft_ext_mask = soft_ext_mask and ext_mask is not None
        if decoder_last_layer is None:
            self.decoder_last_layer = 'softmax' if recon_loss == 'nb' else 'identity'
        else:
            self.decoder_last_layer = decoder_last_layer
        self.use_l_encoder = use_l_encoder
        self.dr_rate = dr_rate
        self.use_dr = self.dr_rate > 0

        if recon_loss == "nb":
            self.theta = torch.nn.Parameter(torch.randn(self.input_dim, self.n_conditions))
        else:
            self.theta = None

        self.hidden_layer_sizes = hidden_layer_sizes
        enc_sizes = [self.input_dim] + list(hidden_layer_sizes)
        self.encoder = ExtEncoder(enc_sizes, self.latent_dim, self.use_bn, self.use_ln, self.use_dr, self.dr_rate,
                                  self.n_conditions, self.n_ext_encoder)

        # soft masks are not applied to the weights: they only steer the L1 proximal operator of the trainer
        if self.soft_mask:
            self.n_inact_genes = (1 - mask).sum().item()
            if mask.shape[0] != latent_dim or mask.shape[1] != input_dim:
                raise ValueError('Incorrect shape of the soft mask.')
            self.mask = mask.t()
            mask = None
        else:
            self.mask = None
        if self.soft_ext_mask:
            self.n_inact_ext_genes = (1 - ext_mask).sum().item()
            if ext_mask.shape[0] != self.n_ext_m_decoder:
                raise ValueError('Dim 0 of ext_mask should be the same as n_ext_m_decoder.')
            if ext_mask.shape[1] != self.input_dim:
                raise ValueError('Dim 1 of ext_mask should be the same as input_dim.')
            self.ext_mask = ext_mask.t()
            ext_mask = None
        else:
            self.ext_mask = None

        self.decoder = MaskedLinearDecoder(self.latent_dim, self.input_dim, self.n_conditions, mask, ext_mask,
                                           self.recon_loss, self.decoder_last_layer, self.n_ext_decoder,
                                           self.n_ext_m_decoder)
        if self.use_l_encoder:
            self.l_encoder = ExtEncoder([self.input_dim, 128], 1, self.use_bn, self.use_ln, self.use_dr, self.dr_rate,
                                        self.n_conditions)
        self._engine = None
        self.seed = 2020

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _hard_masks(self):
        L0 = self.decoder.L0
        m = getattr(L0.expr_L, "mask", None)
        em = getattr(L0.ext_L_m, "mask", None) if self.n_ext_m_decoder else None
        return m, em

    def _spec(self):
        m, em = self._hard_masks()
        return dict(arch="expimap", G=self.input_dim, C=self.n_conditions, H=list(self.hidden_layer_sizes),
                    Z=self.latent_dim, recon=self.recon_loss, use_bn=self.use_bn, use_ln=self.use_ln,
                    dr_rate=self.dr_rate, use_mmd=False, seed=self.seed, n_ext=self.n_ext_decoder,
                    n_ext_m=self.n_ext_m_decoder, mask=m, ext_mask=em, use_hsic=self.use_hsic,
                    hsic_one_vs_all=self.hsic_one_vs_all, hsic_beta=self.hsic_beta,
                    use_l_encoder=self.use_l_encoder, decoder_last_layer=self.decoder_last_layer)

    def _sync_engine_flags(self):
        super()._sync_engine_flags()
        eng = self._engine
        eng.hsic_beta = float(self.hsic_beta)
        # masks are buffers: follow load_state_dict()
        for i, m in enumerate(self._hard_masks()[:len(eng.xp_masks)]):
            if m is not None and eng.xp_masks[i] is not None:
                eng.xp_masks[i].copy_(m.to(eng.device, torch.float32))

    # ------------------------------------------------------------------------------------------ reference API
    def forward(self, x=None, batch=None, sizefactor=None, labeled=None):
        """(recon_loss, kl_div, hsic_loss) for a dense minibatch `x` [B, G] (no autograd graph; the trainer's step
        runs forward + backward + Adam in the engine)."""
        eng, x, B = self._bind_dense(x)
        cond = None if batch is None else batch.to(eng.device).to(torch.int32).contiguous()
        eng.forward_backward(None, cond, B, train=self.training, compute_grad=False, accumulate_logs=False)
        L = eng.losses.clone()
        hs = eng.ws["hsic_raw"][0].clone() if self.use_hsic else torch.zeros((), device=eng.device)
        return L[0], L[1], hs
