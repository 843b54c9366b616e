"""Parameter containers of the expiMap network (never executed - the HIP engine does the arithmetic).

They create the parameters with torch's default initialisers in the construction order of
scarches/models/expimap/modules.py:9-215 (so a torch seed gives the reference's starting weights) and expose its
state-dict names: `decoder.L0.{expr_L,cond_L,ext_L,ext_L_m}.weight`, the buffers `decoder.L0.expr_L.mask` /
`decoder.L0.ext_L_m.mask`, `encoder.expand_{mean,var}_encoder.*`.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn


class MaskedLinear(nn.Linear):
    """Linear whose effective weight is weight * mask; `mask` arrives as [n_in, n_out] and is stored transposed
    ([n_out, n_in], the weight's layout).  Masked weights start at zero."""

    def __init__(self, n_in, n_out, mask, bias=True):
        if n_in != mask.shape[0] or n_out != mask.shape[1]:
            raise ValueError('Incorrect shape of the mask.')
        super().__init__(n_in, n_out, bias)
        self.register_buffer('mask', mask.t())
        self.weight.data *= self.mask


class MaskedCondLayers(nn.Module):
    """expr_L (+ cond_L, ext_L, ext_L_m): input columns are [expr | ext_m | ext | cond]."""

    def __init__(self, n_in: int, n_out: int, n_cond: int, bias: bool, n_ext: int = 0, n_ext_m: int = 0,
                 mask: Optional[torch.Tensor] = None, ext_mask: Optional[torch.Tensor] = None):
        super().__init__()
        self.n_cond, self.n_ext, self.n_ext_m = n_cond, n_ext, n_ext_m
        self.expr_L = nn.Linear(n_in, n_out, bias=bias) if mask is None else MaskedLinear(n_in, n_out, mask, bias=bias)
        if self.n_cond != 0:
            self.cond_L = nn.Linear(self.n_cond, n_out, bias=False)
        if self.n_ext != 0:
            self.ext_L = nn.Linear(self.n_ext, n_out, bias=False)
        if self.n_ext_m != 0:
            self.ext_L_m = (MaskedLinear(self.n_ext_m, n_out, ext_mask, bias=False) if ext_mask is not None
                            else nn.Linear(self.n_ext_m, n_out, bias=False))


class MaskedLinearDecoder(nn.Module):
    def __init__(self, in_dim, out_dim, n_cond, mask, ext_mask, recon_loss, last_layer=None, n_ext=0, n_ext_m=0):
        super().__init__()
        if recon_loss == "mse":
            if last_layer == "softmax":
                raise ValueError("Can't specify softmax last layer with mse loss.")
            last_layer = "identity" if last_layer is None else last_layer
        elif recon_loss == "nb":
            last_layer = "softmax" if last_layer is None else last_layer
        else:
            raise ValueError("Unrecognized loss.")
        if last_layer not in ("softmax", "softplus", "exp", "relu", "identity"):
            raise ValueError("Unrecognized last layer.")
        print("Decoder Architecture:")
        print("\tMasked linear layer in, ext_m, ext, cond, out: ", in_dim, n_ext_m, n_ext, n_cond, out_dim)
        print('\twith hard mask.' if mask is not None else '\twith soft mask.')
        self.n_ext, self.n_ext_m = n_ext, n_ext_m
        self.n_cond = 0 if n_cond is None else n_cond
        self.last_layer = last_layer
        self.L0 = MaskedCondLayers(in_dim, out_dim, n_cond, bias=False, n_ext=n_ext, n_ext_m=n_ext_m, mask=mask,
                                   ext_mask=ext_mask)
        print("Last Decoder layer:", last_layer)

    def nonzero_terms(self):
        """Boolean vector over latent terms: annotated terms whose decoder column is not all zero, plus every
        extension term."""
        v = self.L0.expr_L.weight.data
        nz = (v.norm(p=1, dim=0) > 0).cpu().numpy()
        return np.concatenate([nz, np.full(self.n_ext_m, True), np.full(self.n_ext, True)])

    def n_inactive_terms(self):
        return int((~self.nonzero_terms()).sum())


class ExtEncoder(nn.Module):
    def __init__(self, layer_sizes: list, latent_dim: int, use_bn: bool, use_ln: bool, use_dr: bool, dr_rate: float,
                 num_classes: Optional[int] = None, n_expand: int = 0):
        super().__init__()
        self.n_classes = 0 if num_classes is None else num_classes
        self.n_expand = n_expand
        self.FC = None
        if len(layer_sizes) > 1:
            print("Encoder Architecture:")
            self.FC = nn.Sequential()
            for i, (n_in, n_out) in enumerate(zip(layer_sizes[:-1], layer_sizes[1:])):
                if i == 0:
                    print("\tInput Layer in, out and cond:", n_in, n_out, self.n_classes)
                    self.FC.add_module(f"L{i}", MaskedCondLayers(n_in, n_out, self.n_classes, bias=True))
                else:
                    print("\tHidden Layer", i, "in/out:", n_in, n_out)
                    self.FC.add_module(f"L{i}", nn.Linear(n_in, n_out, bias=True))
                if use_bn:
                    self.FC.add_module(f"N{i}", nn.BatchNorm1d(n_out, affine=True))
                elif use_ln:
                    self.FC.add_module(f"N{i}", nn.LayerNorm(n_out, elementwise_affine=False))
                self.FC.add_module(f"A{i}", nn.ReLU())
                if use_dr:
                    self.FC.add_module(f"D{i}", nn.Dropout(p=dr_rate))
        print("\tMean/Var Layer in/out:", layer_sizes[-1], latent_dim)
        self.mean_encoder = nn.Linear(layer_sizes[-1], latent_dim)
        self.log_var_encoder = nn.Linear(layer_sizes[-1], latent_dim)
        if self.n_expand != 0:
            print("\tExpanded Mean/Var Layer in/out:", layer_sizes[-1], self.n_expand)
            self.expand_mean_encoder = nn.Linear(layer_sizes[-1], self.n_expand)
            self.expand_var_encoder = nn.Linear(layer_sizes[-1], self.n_expand)
