"""sca_out_nb_fused (output-layer GEMM + softmax + NB likelihood + backward in one launch, csrc/out_nb_fused.hip)
against the unfused pair it replaces - sca_gemm_f32 + sca_recon_rows, themselves pinned by the reference's goldens
(tests/test_gpu_ops.py, tests/test_gpu_step.py) - through the C ABI, and against an fp64 evaluation of the reference's
formulas (models/trvae/modules.py:179-181,207, trvae.py:136-141, losses.py:63-102) for the loss and the logit gradient.

Cases: ragged batch sizes (not multiples of the 16-cell workgroup), gene counts that are not multiples of the 16-gene
tile, workgroups that straddle conditions and unsorted conditions, a single condition, empty cells, cells with more
stored counts than the per-cell LDS slots (512), non-integer and large counts (fp64 path), explicit zeros stored in
the CSR matrix, theta outside the tabulated range, compute_grad = 0."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, G, K, C, seed, density=0.15, sorted_cond=True, heavy=0.15, dense_rows=(), empty_rows=(), big_theta=False):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(B, K, generator=g) * 0.7
    W = torch.randn(G, K, generator=g) * (1.5 / K ** 0.5)
    bias = torch.randn(G, generator=g) * 0.3
    theta = torch.randn(G, C, generator=g)
    if big_theta:
        theta[::7] += 6.0                       # exp(theta) > 128: not tabulated, fp64 path
    cond = torch.randint(0, C, (B,), generator=g).to(torch.int32)
    if sorted_cond:
        cond = cond.sort().values
    mask = torch.rand(B, G, generator=g) < density
    for r in dense_rows:
        mask[r] = torch.rand(G, generator=g) < 0.9
    for r in empty_rows:
        mask[r] = False
    x = torch.poisson(torch.rand(B, G, generator=g) * 6.0) + 1.0
    big = torch.rand(B, G, generator=g) < heavy
    x = torch.where(big, x * 9.0 + torch.rand(B, G, generator=g), x)     # counts > 16 and non-integer counts
    x = x * mask
    csr = x.to_sparse_csr()
    indptr, indices, data = csr.crow_indices().clone(), csr.col_indices().to(torch.int32), csr.values().clone()
    if data.numel() > 3:
        data[1] = 0.0                                                    # an explicitly stored zero
        x[0 if indptr[1] > 1 else int((indptr > 1).nonzero()[0]) - 1, indices[1]] = 0.0
    sf = x.sum(1)
    return dict(h=h, W=W, bias=bias, theta=theta, cond=cond, indptr=indptr, indices=indices, data=data, sf=sf, x=x)


def _unfused(lib, _lib, d, B, G, K, C, compute_grad=1):
    dev = lambda t: t.cuda().contiguous()
    h, W, bias, theta = dev(d["h"]), dev(d["W"]), dev(d["bias"]), dev(d["theta"])
    tab_theta, tab_lte = torch.empty(C, G, device="cuda"), torch.empty(C, G, device="cuda")
    _lib.check(lib.sca_theta_prep(theta.data_ptr(), G, C, tab_theta.data_ptr(), tab_lte.data_ptr(), None), "prep")
    logits = torch.empty(B, G, device="cuda")
    _lib.check(lib.sca_gemm_f32(0, 0, h.data_ptr(), K, W.data_ptr(), K, logits.data_ptr(), G, B, G, K, 1, 0,
                                bias.data_ptr(), None), "gemm")
    raw = logits.clone()
    row_ll = torch.empty(B, device="cuda")
    dthetaT, dbias = torch.zeros(C, G, device="cuda"), torch.zeros(G, device="cuda")
    ip, ix, dv = dev(d["indptr"]), dev(d["indices"]), dev(d["data"])
    sf, cond = dev(d["sf"]), dev(d["cond"])
    _lib.check(lib.sca_recon_rows(0, logits.data_ptr(), G, None, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), None,
                                  sf.data_ptr(), cond.data_ptr(), tab_theta.data_ptr(), tab_lte.data_ptr(), B, G, C,
                                  4, row_ll.data_ptr(), dthetaT.data_ptr(), dbias.data_ptr(), None, compute_grad,
                                  None, None, None), "recon_rows")
    torch.cuda.synchronize()
    return dict(dl=logits, row_ll=row_ll, dthetaT=dthetaT, dbias=dbias, raw=raw, tab_theta=tab_theta,
                dev=(h, W, bias, ip, ix, dv, sf, cond))


def _fused(lib, _lib, d, B, G, K, C, tab_theta, devs, compute_grad=1, row_ids=None):
    h, W, bias, ip, ix, dv, sf, cond = devs
    gam = torch.empty(C, G, 17, 2, device="cuda")
    _lib.check(lib.sca_nb_gamma_table(tab_theta.data_ptr(), C, G, gam.data_ptr(), None), "gamma_table")
    dl = torch.full((B, G), float("nan"), device="cuda")
    ovf = torch.full((B, G), float("nan"), device="cuda")
    row_ll = torch.full((B,), float("nan"), device="cuda")
    dthetaT, dbias = torch.zeros(C, G, device="cuda"), torch.zeros(G, device="cuda")
    _lib.check(lib.sca_out_nb_fused(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), ip.data_ptr(), ix.data_ptr(),
                                    dv.data_ptr(), None if row_ids is None else row_ids.data_ptr(), sf.data_ptr(),
                                    cond.data_ptr(), tab_theta.data_ptr(), gam.data_ptr(), B, G, K, C, dl.data_ptr(), G,
                                    ovf.data_ptr(), G, row_ll.data_ptr(), dthetaT.data_ptr(), dbias.data_ptr(),
                                    compute_grad, None), "out_nb_fused")
    torch.cuda.synchronize()
    return dict(dl=dl, row_ll=row_ll, dthetaT=dthetaT, dbias=dbias)


def _ref64(d, B):
    """The reference's formulas in fp64 with autograd: row log-likelihoods and d(-sum ll / B) / d logits."""
    h, W, bias = d["h"].double(), d["W"].double(), d["bias"].double()
    logits = (h @ W.t() + bias).requires_grad_(True)
    gamma = torch.softmax(logits, -1)
    mu = gamma * d["sf"].double()[:, None]
    par = d["theta"].double().clone().requires_grad_(True)
    th = torch.exp(par.t()[d["cond"].long()])
    x, eps = d["x"].double(), 1e-8
    lt = torch.log(th + mu + eps)
    ll = (th * (torch.log(th + eps) - lt) + x * (torch.log(mu + eps) - lt) + torch.lgamma(x + th) - torch.lgamma(th)
          - torch.lgamma(x + 1))
    row = ll.sum(1)
    (-row.sum() / B).backward()
    # dthetaT of the kernels = sum over the condition's cells of theta dll/dtheta = -B * d loss / d theta-parameter
    return row.detach(), logits.grad, (-B * par.grad).t().contiguous()


CASES = [
    dict(B=16, G=64, K=32, C=1, seed=1),
    dict(B=37, G=200, K=32, C=3, seed=2, dense_rows=(5,), empty_rows=(0, 36)),
    dict(B=128, G=1000, K=128, C=5, seed=3, sorted_cond=False),
    dict(B=250, G=2000, K=128, C=20, seed=4, dense_rows=(17, 18, 249), empty_rows=(100,)),
    dict(B=96, G=2048, K=64, C=4, seed=5, big_theta=True),
    dict(B=1, G=17, K=32, C=2, seed=6),
    dict(B=40, G=300, K=256, C=2, seed=7),
]


@pytest.mark.parametrize("cs", CASES, ids=lambda c: "B%d_G%d_K%d_C%d" % (c["B"], c["G"], c["K"], c["C"]))
def test_out_nb_fused_matches_unfused_and_fp64(cs):
    from scarches_amd import _lib
    lib = _lib.load()
    cs = dict(cs)
    B, G, K, C = cs.pop("B"), cs.pop("G"), cs.pop("K"), cs.pop("C")
    assert lib.sca_out_nb_fused_supported(G, K) == 1
    d = _case(B, G, K, C, **cs)
    u = _unfused(lib, _lib, d, B, G, K, C)
    f = _fused(lib, _lib, d, B, G, K, C, u["tab_theta"], u["dev"])
    row64, dl64, dth64 = _ref64(d, B)
    # against fp64: the two HIP paths must be equally close to the truth
    sc_ll = row64.abs().max().item()
    sc_dl = dl64.abs().max().item()
    for name, o in (("unfused", u), ("fused", f)):
        assert (o["row_ll"].cpu().double() - row64).abs().max().item() <= 2e-5 * sc_ll, name
        assert (o["dl"].cpu().double() - dl64).abs().max().item() <= 2e-5 * sc_dl, name
    assert torch.isfinite(f["dl"]).all() and torch.isfinite(f["row_ll"]).all()
    # against the unfused pair: fp32 rounding only
    for k in ("row_ll", "dl", "dbias"):
        ref = u[k]
        scale = ref.abs().max().item() + 1e-30
        err = (f[k] - ref).abs().max().item()
        assert err <= 2e-5 * scale, (k, err, scale)
    # theta gradient: theta (theta / (theta + eps) - theta / (theta + mu + eps)) cancels in fp32 for large theta (the
    # big_theta case: exp(theta) ~ 1e3 - 1e4) in BOTH kernels; the yardstick is the fp64 value and the unfused kernel's
    # own distance from it
    sc = dth64.abs().max().item()
    e_u = (u["dthetaT"].cpu().double() - dth64).abs().max().item()
    e_f = (f["dthetaT"].cpu().double() - dth64).abs().max().item()
    assert e_f <= 2e-5 * sc + 2.0 * e_u, (e_f, e_u, sc)
    # loss only: same row sums, nothing else touched
    f0 = _fused(lib, _lib, d, B, G, K, C, u["tab_theta"], u["dev"], compute_grad=0)
    assert torch.allclose(f0["row_ll"], f["row_ll"], rtol=1e-6, atol=1e-6 * sc_ll)
    assert float(f0["dthetaT"].abs().max()) == 0.0 and float(f0["dbias"].abs().max()) == 0.0


def test_out_nb_fused_row_ids_and_rejections():
    from scarches_amd import _lib
    lib = _lib.load()
    B, G, K, C = 48, 300, 32, 3
    d = _case(B, G, K, C, seed=9)
    u = _unfused(lib, _lib, d, B, G, K, C)
    # the same cells addressed through row ids of a larger matrix (the CSR of the whole data set, rows permuted)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(B)
    xs = d["x"][perm].to_sparse_csr()
    devs = list(u["dev"])
    devs[3], devs[4], devs[5] = (xs.crow_indices().cuda(), xs.col_indices().to(torch.int32).cuda(), xs.values().cuda())
    # (the stored explicit zero of _case is gone in this re-built CSR: values are identical otherwise)
    f = _fused(lib, _lib, d, B, G, K, C, u["tab_theta"], tuple(devs), row_ids=inv.cuda())
    for k in ("row_ll", "dl", "dthetaT", "dbias"):
        scale = u[k].abs().max().item() + 1e-30
        assert (f[k] - u[k]).abs().max().item() <= 2e-5 * scale, k
    assert lib.sca_out_nb_fused_supported(4000, 128) == 0 and lib.sca_out_nb_fused_supported(2000, 100) == 0
    assert lib.sca_out_nb_fused_supported(2000, 16) == 0
    h = u["dev"][0]
    args = [h.data_ptr(), K, u["dev"][1].data_ptr(), K, u["dev"][2].data_ptr(), None, None, None, None, None, None,
            u["tab_theta"].data_ptr(), u["tab_theta"].data_ptr(), B, 4000, K, C, f["dl"].data_ptr(), 4000, None, 0,
            f["row_ll"].data_ptr(), None, None, 0, None]
    assert lib.sca_out_nb_fused(*args) == -2


def test_recon_rows_with_gamma_table_is_identical():
    """sca_recon_rows_tab with the per-step table of sca_nb_gamma_table against sca_recon_rows: the table holds the
    results of the same per-count loop, so losses and gradients agree to fp32 rounding."""
    from scarches_amd import _lib
    lib = _lib.load()
    B, G, K, C = 200, 700, 32, 6
    d = _case(B, G, K, C, seed=21, big_theta=True)
    u = _unfused(lib, _lib, d, B, G, K, C)
    h, W, bias, ip, ix, dv, sf, cond = u["dev"]
    gam = torch.empty(C, G, 17, 2, device="cuda")
    _lib.check(lib.sca_nb_gamma_table(u["tab_theta"].data_ptr(), C, G, gam.data_ptr(), None), "gamma_table")
    tab_lte = torch.log(u["tab_theta"] + 1e-8)
    logits = u["raw"].clone()
    row_ll = torch.empty(B, device="cuda")
    dthetaT, dbias = torch.zeros(C, G, device="cuda"), torch.zeros(G, device="cuda")
    _lib.check(lib.sca_recon_rows_tab(0, logits.data_ptr(), G, None, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), None,
                                      sf.data_ptr(), cond.data_ptr(), u["tab_theta"].data_ptr(), tab_lte.data_ptr(), B,
                                      G, C, 4, row_ll.data_ptr(), dthetaT.data_ptr(), dbias.data_ptr(), None, 1, None,
                                      None, gam.data_ptr(), None), "recon_rows_tab")
    torch.cuda.synchronize()
    # (same formulas; the two loops are compiled separately, so products may be contracted differently: last-bit noise)
    for k, v in (("row_ll", row_ll), ("dl", logits), ("dthetaT", dthetaT), ("dbias", dbias)):
        assert (v - u[k]).abs().max().item() <= 2e-6 * u[k].abs().max().item(), k
