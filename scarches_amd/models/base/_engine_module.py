"""Shared plumbing of the network objects (`trVAE`, `expiMap`, `scpoli`): the nn.Module keeps the reference's
parameter names / state dict, the HIP engine owns the storage and does the arithmetic."""
from collections import OrderedDict
from typing import Optional

import torch

from ... import _lib
from ...engine import CVAEEngine


class EngineBackedModule(torch.nn.Module):
    """nn.Module whose parameters become views of a CVAEEngine's flat HBM buffers on first use."""

    _engine: Optional[CVAEEngine] = None
    seed = 2020

    def _spec(self) -> dict:
        raise NotImplementedError

    def engine(self, device=None) -> CVAEEngine:
        """The HIP engine holding this network's parameters (created on first use; parameters then become views of
        the engine's flat HBM buffer so state_dict()/load_state_dict()/requires_grad keep working on the real data)."""
        if self._engine is None:
            if not torch.cuda.is_available():
                raise _lib.HipLibraryError("no HIP device: scarches_amd has no CPU fallback")
            if device is None:
                p = next(self.parameters())
                device = p.device if p.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
            named = OrderedDict((k, p.detach()) for k, p in self.named_parameters())
            spec = dict(self._spec())
            spec["buffers"] = {k: b for k, b in self.named_buffers()}       # BatchNorm running statistics
            spec["freeze"] = bool(getattr(self, "freeze", False))
            eng = CVAEEngine(spec, named, device)
            for k, p in self.named_parameters():
                p.data = eng.p[k]
            for k, b in self.named_buffers():       # running statistics are updated in place by the engine
                if k in eng.buffers:
                    b.data = eng.buffers[k]
            self._engine = eng
        self._sync_engine_flags()
        return self._engine

    def _sync_engine_flags(self):
        self._engine.set_trainable([k for k, p in self.named_parameters() if p.requires_grad])
        self._engine.freeze = bool(getattr(self, "freeze", False))

    def _apply(self, fn, *args, **kwargs):
        # .cuda() / .to(device): moving would detach the parameters from the flat buffer, so rebuild the engine lazily
        if self._engine is not None:
            for k, p in self.named_parameters():
                p.data = p.data.clone()
            self._engine = None
        return super()._apply(fn, *args, **kwargs)

    def sampling(self, mu, log_var):
        """z = mu + eps * sqrt(exp(log_var) + 1e-4), eps ~ N(0, 1) (models/base/_base.py:391-407)."""
        var = torch.exp(log_var) + 1e-4
        return mu + torch.randn_like(mu) * var.sqrt()

    def _bind_dense(self, x):
        """Upload a dense minibatch as the engine's CSR matrix; returns (engine, x on device, B)."""
        eng = self.engine(x.device if x.is_cuda else None)
        x = x.to(eng.device, torch.float32).contiguous()
        csr = x.to_sparse_csr()
        eng.bind_csr(csr.crow_indices().to(torch.int64), csr.col_indices().to(torch.int32), csr.values())
        return eng, x, x.shape[0]

    def get_latent(self, x, c=None, mean=False, mean_var=False):
        eng = self.engine(x.device if x.is_cuda else None)
        x = x.to(eng.device, torch.float32).contiguous()
        cond = None if c is None else c.to(eng.device).to(torch.int32).contiguous()
        z, mu, var = eng.encode(None, cond, x.shape[0], mean=False, x_dense=x)
        if mean:
            return mu.clone()
        if mean_var:
            return mu.clone(), var.clone()
        return z.clone()
