"""`trVAE`: the network object behind `TRVAE.model`, API of scarches/models/trvae/trvae.py:14-157.

Same constructor, attributes (`condition_encoder`, `freeze`, `use_mmd`, ...), parameter names and
`forward(x, batch, sizefactor, labeled) -> (recon_loss, kl_div, mmd_loss)`; the arithmetic is executed by the HIP
engine.  Gradients are NOT produced by autograd: `train_step()` runs forward + hand-written backward + Adam in the
engine and is what the trainer calls; `forward()` returns detached 0-d loss tensors.
"""
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from ..base._engine_module import EngineBackedModule
from .modules import Decoder, Encoder


class trVAE(EngineBackedModule):
    def __init__(self,
                 input_dim: int,
                 conditions: list,
                 hidden_layer_sizes: list = [256, 64],
                 latent_dim: int = 10,
                 dr_rate: float = 0.05,
                 use_mmd: bool = False,
                 mmd_on: str = 'z',
                 mmd_boundary: Optional[int] = None,
                 recon_loss: Optional[str] = 'nb',
                 beta: float = 1,
                 use_bn: bool = False,
                 use_ln: bool = True,
                 ):
        super().__init__()
        assert isinstance(hidden_layer_sizes, list)
        assert isinstance(latent_dim, int)
        assert isinstance(conditions, list)
        assert recon_loss in ["mse", "nb", "zinb"], "'recon_loss' must be 'mse', 'nb' or 'zinb'"

        print("\nINITIALIZING NEW NETWORK..............")
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.n_conditions = len(conditions)
        self.conditions = conditions
        self.condition_encoder = {k: v for k, v in zip(conditions, range(len(conditions)))}
        self.cell_type_encoder = None
        self.recon_loss = recon_loss
        self.mmd_boundary = mmd_boundary
        self.use_mmd = use_mmd
        self.freeze = False
        self.beta = beta
        self.use_bn = use_bn
        self.use_ln = use_ln
        self.mmd_on = mmd_on
        self.dr_rate = dr_rate
        self.use_dr = self.dr_rate > 0

        if recon_loss in ["nb", "zinb"]:
            self.theta = torch.nn.Parameter(torch.randn(self.input_dim, self.n_conditions))
        else:
            self.theta = None

        self.hidden_layer_sizes = hidden_layer_sizes
        enc_sizes = [self.input_dim] + list(hidden_layer_sizes)
        dec_sizes = list(reversed(hidden_layer_sizes)) + [self.input_dim]
        self.encoder = Encoder(enc_sizes, self.latent_dim, self.use_bn, self.use_ln, self.use_dr, self.dr_rate,
                               self.n_conditions)
        self.decoder = Decoder(dec_sizes, self.latent_dim, self.recon_loss, self.use_bn, self.use_ln, self.use_dr,
                               self.dr_rate, self.n_conditions)
        self._engine = None
        self.seed = 2020

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _spec(self):
        return dict(G=self.input_dim, C=self.n_conditions, H=list(self.hidden_layer_sizes), Z=self.latent_dim,
                    recon=self.recon_loss, use_bn=self.use_bn, use_ln=self.use_ln, dr_rate=self.dr_rate,
                    use_mmd=self.use_mmd, mmd_on=self.mmd_on, boundary=self.mmd_boundary, beta=float(self.beta),
                    seed=self.seed)

    def _sync_engine_flags(self):
        super()._sync_engine_flags()
        eng = self._engine
        eng.use_mmd, eng.beta, eng.boundary, eng.mmd_on = self.use_mmd, float(self.beta), self.mmd_boundary, self.mmd_on

    # ------------------------------------------------------------------------------------------ reference API
    def forward(self, x=None, batch=None, sizefactor=None, labeled=None):
        """(recon_loss, kl_div, mmd_loss) for a dense minibatch `x` [B, G] on the HIP device (no autograd graph).
        Train mode samples dropout masks / z like the reference; sizefactor is recomputed from x (== x.sum(1))."""
        eng, x, B = self._bind_dense(x)
        cond = None if batch is None else batch.to(eng.device).to(torch.int32).contiguous()
        eng.forward_backward(None, cond, B, train=self.training, compute_grad=False, accumulate_logs=False)
        L = eng.losses.clone()
        return L[0], L[1], L[2]

    def get_y(self, x, c=None):
        eng = self.engine(x.device if x.is_cuda else None)
        x = x.to(eng.device, torch.float32).contiguous()
        cond = None if c is None else c.to(eng.device).to(torch.int32).contiguous()
        eng.encode(None, cond, x.shape[0], mean=False, x_dense=x)
        return eng.decode_first_layer(cond, x.shape[0]).clone()
