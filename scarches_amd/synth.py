"""Synthetic negative-binomial count matrices for the benchmark and the scale tests.

    x_bg ~ NB(mean = s_b * pi[g, t(b)] * exp(delta[g, c(b)]), theta_g)
    pi: softmax of N(0, 1.5^2) logits per latent cell type (T = 12), delta ~ N(0, 0.3^2) condition shift,
    s_b ~ LogNormal(log(lib), 0.5), theta_g ~ Gamma(2, 1); conditions with unequal sizes (Dirichlet(2)).

Generated in chunks on the given device (torch is used as an RNG / allocator here, not as a compute path of the
product) and returned as a canonical CSR triplet that stays resident in HBM.
"""
from __future__ import annotations

import torch


def make_synthetic_csr(n_cells: int, n_genes: int, n_cond: int, device, seed: int = 0, n_types: int = 12,
                       lib: float = 3000.0, chunk: int = 65536, density_target: float = 0.10):
    """Returns (indptr int64 [N+1], indices int32 [nnz], data float32 [nnz], cond int32 [N], cell_type int32 [N])."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    logits = torch.randn(n_types, n_genes, generator=g, device=dev) * 1.5
    # a heavy-tailed expression profile keeps most genes silent per cell (5-15 % density at lib ~ 3000)
    pi = torch.softmax(logits * 2.0, dim=1)
    delta = torch.randn(n_cond, n_genes, generator=g, device=dev) * 0.3
    theta = torch._standard_gamma(torch.full((n_genes,), 2.0, device=dev), generator=g) + 0.05
    p_cond = torch._standard_gamma(torch.full((n_cond,), 2.0, device=dev), generator=g)
    p_cond = p_cond / p_cond.sum()
    cond = torch.multinomial(p_cond, n_cells, replacement=True, generator=g).to(torch.int32)
    ctype = torch.randint(0, n_types, (n_cells,), generator=g, device=dev).to(torch.int32)
    scale = lib * (density_target / 0.10)
    counts, idx_l, dat_l = [], [], []
    for s in range(0, n_cells, chunk):
        e = min(n_cells, s + chunk)
        c, t = cond[s:e].long(), ctype[s:e].long()
        sz = scale * torch.exp(torch.randn(e - s, generator=g, device=dev) * 0.5)
        mean = sz[:, None] * pi[t] * torch.exp(delta[c])
        lam = torch._standard_gamma(theta[None, :].expand_as(mean).contiguous(), generator=g) * (mean / theta[None, :])
        x = torch.poisson(lam, generator=g)
        nz = x > 0
        counts.append(nz.sum(1))
        r, cidx = nz.nonzero(as_tuple=True)
        idx_l.append(cidx.to(torch.int32))
        dat_l.append(x[r, cidx].to(torch.float32))
        del x, lam, mean, nz
    counts = torch.cat(counts)
    indptr = torch.zeros(n_cells + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(counts, 0)
    return indptr, torch.cat(idx_l), torch.cat(dat_l), cond, ctype
