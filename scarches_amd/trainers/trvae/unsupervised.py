import torch

from .trainer import Trainer


class trVAETrainer(Trainer):
    """Unsupervised trVAE / CVAE trainer (API of scarches/trainers/trvae/unsupervised.py:5-71): all arguments are
    those of `Trainer`; the loss is recon + alpha(epoch, iter) * kl + mmd."""

    def __init__(self, model, adata, **kwargs):
        super().__init__(model, adata, **kwargs)

    def loss(self, total_batch=None):
        row_ids, cond = total_batch[0], total_batch[1]
        # engine.losses = [recon, kl, mmd, recon + alpha*kl + mmd]; gradients land in engine.flat_g
        return self.engine.forward_backward(row_ids, cond, row_ids.numel(), train=True, compute_grad=True)
