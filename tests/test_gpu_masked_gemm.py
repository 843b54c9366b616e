"""Tile-skipping masked decoder kernels (csrc/masked_gemm.hip) through the C ABI against plain fp64 matrix products of
the masked weight (MaskedLinear, models/expimap/modules.py:9-25: logits = z @ (W o M)^T and its autograd), at ragged
shapes (G not a multiple of 32, B not a multiple of the row tiles, odd active-column counts, empty gene tiles) and at
the cfg3 shape; and the expiMap goldens of the real reference pushed through that path.

Tolerance: fp32 dot products of length K: |got - ref64| <= 1e-5 * max|ref64| (the engine-level tests apply the
reference-noise yardstick of tests/_util.grads_close on top)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests._util import golden_ref64, grads_close

pytestmark = pytest.mark.gpu


def plan(pat):
    """The caller-side plan of include/scarches_hip.h (same construction as CVAEEngine._masked_meta)."""
    G, Zt = pat.shape
    NT = (G + 31) // 32
    pad = np.zeros((NT * 32, Zt), bool)
    pad[:G] = pat
    active = pad.reshape(NT, 32, Zt).any(1)
    k = (active.sum(1) + 1) & ~1
    Kmax = max((int(k.max()) + 7) & ~7, 8)
    cols = np.full((NT, Kmax), Zt, np.int32)
    for t in range(NT):
        c = np.nonzero(active[t])[0]
        cols[t, :len(c)] = c
    return NT, Kmax, torch.tensor(k.astype(np.int32)).cuda(), torch.tensor(cols).cuda()


@pytest.mark.parametrize("B,G,Zt,density", [(200, 77, 13, 0.2), (37, 33, 300, 0.02), (1000, 5000, 280, 0.012),
                                            (129, 96, 40, 0.0), (4096, 2000, 64, 0.05)])
def test_masked_products_match_dense_fp64(B, G, Zt, density):
    from scarches_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(B + G)
    pat = rng.random((G, Zt)) < density
    if G > 64:
        pat[32:64] = False                      # a gene tile with no active program at all
    if density == 0.0:
        pat[5, 7] = True
    W = torch.tensor(rng.normal(size=(G, Zt)).astype(np.float32))
    Weff = (W * torch.tensor(pat)).cuda().contiguous()
    z = torch.tensor(rng.normal(size=(B, Zt)).astype(np.float32)).cuda()
    dl = torch.tensor(rng.normal(size=(B, G)).astype(np.float32)).cuda()
    NT, Kmax, tk, tc = plan(pat)
    assert Kmax <= L.sca_masked_max_cols()
    meta = (tk.data_ptr(), tc.data_ptr(), NT, Kmax)
    st = torch.cuda.current_stream().cuda_stream
    n = NT * Kmax * 32
    Wc = torch.full((2 * n,), float("nan"), device="cuda")
    _lib.check(L.sca_masked_pack(Weff.data_ptr(), Zt, *meta, G, Zt, Wc.data_ptr(), st), "pack")
    # forward
    out = torch.full((B, G), float("nan"), device="cuda")
    _lib.check(L.sca_masked_fwd(z.data_ptr(), Zt, Wc.data_ptr(), *meta, B, G, Zt, out.data_ptr(), G, st), "fwd")
    ref = z.double() @ Weff.double().T
    assert float((out.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-30)
    # dX
    S = L.sca_masked_dx_slabs(B, NT)
    slabs = torch.full((S, B, Zt), float("nan"), device="cuda")
    _lib.check(L.sca_masked_dx(dl.data_ptr(), G, Wc.data_ptr(), *meta, B, G, Zt, slabs.data_ptr(), B * Zt, st), "dx")
    ref = dl.double() @ Weff.double()
    assert float((slabs.double().sum(0) - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-30)
    # dW (+ scatter back; masked-out entries of an active column are left to the mask multiply of the caller)
    S = L.sca_masked_dw_slabs(B, NT)
    dWc = torch.full((S, n), float("nan"), device="cuda")
    _lib.check(L.sca_masked_dw(z.data_ptr(), Zt, dl.data_ptr(), G, *meta, B, G, Zt, dWc.data_ptr(), n, st), "dw")
    dWeff = torch.zeros(G, Zt, device="cuda")
    _lib.check(L.sca_masked_unpack(dWc.data_ptr(), S, n, *meta, G, Zt, dWeff.data_ptr(), Zt, st), "unpack")
    ref = dl.double().T @ z.double()
    patd = torch.tensor(pat).cuda()
    got = dWeff.double() * patd
    assert torch.isfinite(dWeff).all()
    assert float((got - ref * patd).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-30)
    torch.cuda.synchronize()


def test_unsupported_plans_are_refused():
    from scarches_amd import _lib
    L = _lib.load()
    G, Zt = 64, 400
    pat = np.ones((G, Zt), bool)
    NT, Kmax, tk, tc = plan(pat)
    x = torch.zeros(8, device="cuda")
    assert Kmax > L.sca_masked_max_cols()
    assert L.sca_masked_pack(x.data_ptr(), Zt, tk.data_ptr(), tc.data_ptr(), NT, Kmax, G, Zt, x.data_ptr(), None) == -2
    assert L.sca_masked_fwd(x.data_ptr(), Zt, x.data_ptr(), tk.data_ptr(), tc.data_ptr(), NT + 1, 8, 8, G, Zt,
                            x.data_ptr(), G, None) == -1


@pytest.mark.parametrize("name", ["expimap_nb_hard", "expimap_nb_ext", "expimap_nb_hsic_ova", "expimap_nb_softplus",
                                  "expimap_nb_lenc"])
def test_expimap_goldens_through_the_tile_skipping_path(name, monkeypatch):
    """The real reference's losses and gradients for the hard-masked goldens, with the decoder forced onto the
    tile-skipping kernels (their masks are too dense for the engine to choose that path by itself)."""
    monkeypatch.setenv("SCA_MASKED_MIN_SKIP", "-1")
    from tests.test_gpu_expimap import make_engine, step
    eng, z, cfg, W, Gr, masks = make_engine(name)
    assert eng.masked is not None and not cfg["soft_mask"]
    eng.set_hyper(alpha_kl=float(z["alpha"]))
    step(eng, z, cfg, masks)
    torch.cuda.synchronize()
    Lh = eng.losses.cpu().numpy()
    recon, kl = float(z["recon"]), float(z["kl"])
    assert abs(Lh[0] - recon) <= 1e-5 * abs(recon) and abs(Lh[1] - kl) <= 1e-5 * abs(kl)
    got = {k: eng.g[k].cpu().numpy() for k in eng.names}
    bad = grads_close(got, Gr, rtol=1e-5, floor_frac=1e-7, ref64=golden_ref64(name)[1], tag=name)
    assert not bad, bad
    assert np.all(got["decoder.L0.expr_L.weight"][z["mask"].T == 0] == 0)
    # Adam trajectory of the fixture through the same path
    if "adam_steps" in z.files and cfg["alpha_gl"] is None and cfg["alpha_l1"] is None and cfg["gamma_ext"] is None:
        eng2, z, cfg, W, Gr, masks = make_engine(name)
        for _ in range(int(z["adam_steps"])):
            eng2.begin_step(alpha_kl=float(z["alpha"]), lr=1e-3, eps=0.01, clip=0.0)
            step(eng2, z, cfg, masks)
            eng2.adam_step(weight_decay=float(z["adam_wd"]))
        torch.cuda.synchronize()
        for k in eng2.names:
            ref = z["adam/" + k]
            assert np.max(np.abs(eng2.p[k].cpu().numpy() - ref)) <= 4e-6 * max(1.0, np.max(np.abs(ref))), k
