"""Parameter containers of the trVAE network.

These nn.Modules exist to (1) create the parameters with torch's default initialisers in the same order as the
reference (scarches/models/trvae/modules.py:7-208), so a given torch seed yields the same starting weights, and
(2) expose the reference's state-dict names (`encoder.FC.L0.expr_L.weight`, `decoder.FirstL.L0.cond_L.weight`,
`decoder.mean_decoder.0.bias`, ...) on which checkpoints, surgery expansion and the freeze rules depend.
They are never executed: all arithmetic runs in the HIP engine (scarches_amd/engine.py).
"""
import torch.nn as nn


class CondLayers(nn.Module):
    """expr_L: Linear(n_in -> n_out, bias); cond_L: Linear(n_cond -> n_out, no bias) when n_cond > 0."""

    def __init__(self, n_in: int, n_out: int, n_cond: int, bias: bool):
        super().__init__()
        self.n_cond = n_cond
        self.expr_L = nn.Linear(n_in, n_out, bias=bias)
        if self.n_cond != 0:
            self.cond_L = nn.Linear(self.n_cond, n_out, bias=False)


def _norm(name, seq, width, use_bn, use_ln):
    if use_bn:
        seq.add_module(name, nn.BatchNorm1d(width, affine=True))
    elif use_ln:
        seq.add_module(name, nn.LayerNorm(width, elementwise_affine=False))


class Encoder(nn.Module):
    def __init__(self, layer_sizes, latent_dim, use_bn, use_ln, use_dr, dr_rate, num_classes=None):
        super().__init__()
        self.n_classes = 0 if num_classes is None else num_classes
        self.FC = None
        if len(layer_sizes) > 1:
            print("Encoder Architecture:")
            self.FC = nn.Sequential()
            for i, (n_in, n_out) in enumerate(zip(layer_sizes[:-1], layer_sizes[1:])):
                if i == 0:
                    print("\tInput Layer in, out and cond:", n_in, n_out, self.n_classes)
                    self.FC.add_module(f"L{i}", CondLayers(n_in, n_out, self.n_classes, bias=True))
                else:
                    print("\tHidden Layer", i, "in/out:", n_in, n_out)
                    self.FC.add_module(f"L{i}", nn.Linear(n_in, n_out, bias=True))
                _norm(f"N{i}", self.FC, n_out, use_bn, use_ln)
                self.FC.add_module(f"A{i}", nn.ReLU())
                if use_dr:
                    self.FC.add_module(f"D{i}", nn.Dropout(p=dr_rate))
        print("\tMean/Var Layer in/out:", layer_sizes[-1], latent_dim)
        self.mean_encoder = nn.Linear(layer_sizes[-1], latent_dim)
        self.log_var_encoder = nn.Linear(layer_sizes[-1], latent_dim)


class Decoder(nn.Module):
    def __init__(self, layer_sizes, latent_dim, recon_loss, use_bn, use_ln, use_dr, dr_rate, num_classes=None):
        super().__init__()
        self.use_dr = use_dr
        self.recon_loss = recon_loss
        self.n_classes = 0 if num_classes is None else num_classes
        sizes = [latent_dim] + list(layer_sizes)
        print("Decoder Architecture:")
        print("\tFirst Layer in, out and cond: ", sizes[0], sizes[1], self.n_classes)
        self.FirstL = nn.Sequential()
        self.FirstL.add_module("L0", CondLayers(sizes[0], sizes[1], self.n_classes, bias=False))
        _norm("N0", self.FirstL, sizes[1], use_bn, use_ln)
        self.FirstL.add_module("A0", nn.ReLU())
        if use_dr:
            self.FirstL.add_module("D0", nn.Dropout(p=dr_rate))
        if len(sizes) > 2:
            self.HiddenL = nn.Sequential()
            for i, (n_in, n_out) in enumerate(zip(sizes[1:-1], sizes[2:])):
                if i + 3 < len(sizes):
                    print("\tHidden Layer", i + 1, "in/out:", n_in, n_out)
                    self.HiddenL.add_module(f"L{i + 1}", nn.Linear(n_in, n_out, bias=False))
                    _norm(f"N{i + 1}", self.HiddenL, n_out, use_bn, use_ln)
                    self.HiddenL.add_module(f"A{i + 1}", nn.ReLU())
                    if use_dr:
                        self.HiddenL.add_module(f"D{i + 1}", nn.Dropout(p=dr_rate))
        else:
            self.HiddenL = None
        print("\tOutput Layer in/out: ", sizes[-2], sizes[-1], "\n")
        if recon_loss == "mse":
            self.recon_decoder = nn.Sequential(nn.Linear(sizes[-2], sizes[-1]), nn.ReLU())
        if recon_loss == "poisson":      # scPoli's copy of the decoder only (models/scpoli/scpoli.py:607-609)
            self.recon_decoder = nn.Sequential(nn.Linear(sizes[-2], sizes[-1]), nn.Softmax(dim=-1))
        if recon_loss == "zinb":
            self.mean_decoder = nn.Sequential(nn.Linear(sizes[-2], sizes[-1]), nn.Softmax(dim=-1))
            self.dropout_decoder = nn.Linear(sizes[-2], sizes[-1])
        if recon_loss == "nb":
            self.mean_decoder = nn.Sequential(nn.Linear(sizes[-2], sizes[-1]), nn.Softmax(dim=-1))
