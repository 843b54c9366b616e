"""GPU parity of the individual HIP entry points (through the C ABI) against the oracle / golden vectors."""
import json

import numpy as np
import pytest
import torch

from tests._util import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from scarches_amd import _lib
    return _lib.load()


_KEEP = []


def dev(a, dtype=None):
    """Host array -> device tensor that stays alive until the end of the test (see _release)."""
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    t = t.cuda().contiguous()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 130, 77), (16, 2000, 128), (4096, 128, 2000),
                                   (7, 5, 3), (512, 20, 128), (300, 10, 128)])
@pytest.mark.parametrize("mode", [(0, 0), (0, 1), (1, 1)])
def test_gemm_modes(lib, M, N, K, mode):
    from scarches_amd import _lib
    am, bm = mode
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    ref = (A.double() @ Bm.double())
    Ad = dev(A if am == 0 else A.t())           # mode 1 stores A as [K][M]
    Bd = dev(Bm.t() if bm == 0 else Bm)         # mode 0 stores B as [N][K]
    lda = K if am == 0 else M
    ldb = K if bm == 0 else N
    for S in sorted({1, lib.sca_gemm_splitk_heuristic(M, N, K), 3}):
        slabs = torch.full((S, M, N), float("nan"), device="cuda")
        _lib.check(lib.sca_gemm_f32(am, bm, Ad.data_ptr(), lda, Bd.data_ptr(), ldb, slabs.data_ptr(), N, M, N, K, S,
                                    M * N, None, st()), "gemm")
        out = torch.empty(M, N, device="cuda")
        _lib.check(lib.sca_splitk_reduce(slabs.data_ptr(), S, M * N, out.data_ptr(), M * N, st()), "reduce")
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs().max().item()
        scale = (A.abs().double() @ Bm.abs().double()).max().item()
        assert err <= 2e-6 * scale, (S, err, scale)


@pytest.mark.parametrize("M", [4, 132, 2000])
@pytest.mark.parametrize("N", [2, 66, 130])
@pytest.mark.parametrize("K", [17, 1000, 8192 + 3])
def test_gemm_kk_shapes(lib, M, N, K):
    """sca_gemm_kk_f32 (register-direct C = A^T B, both operands k-major) against an fp64 product: clamped lanes
    (M, N not multiples of the 128 x 64 wave tile), k tails, waves whose k range is empty (K = 17 with 3 x 4 wave
    ranges), split counts 1 / 3 / heuristic with their slab strides, padded leading dimensions and an output pitch that
    is not a multiple of 4 (scalar store path)."""
    from scarches_amd import _lib
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    lda, ldb = M + 4, N + 2                        # padded operand pitches (multiples of 4 / 2)
    A = torch.randn(K, lda, generator=g)
    Bm = torch.randn(K, ldb, generator=g)
    ref = A[:, :M].double().t() @ Bm[:, :N].double()
    scale = (A[:, :M].abs().double().t() @ Bm[:, :N].abs().double()).max().item()
    Ad, Bd = dev(A), dev(Bm)
    assert lib.sca_gemm_kk_supported(Ad.data_ptr(), lda, Bd.data_ptr(), ldb, M, N) == 1
    for S in sorted({1, 3, lib.sca_gemm_kk_splitk_heuristic(M, N, K)}):
        for ldc in (N, N + 3):                     # N + 3: rows of C that are not 16-byte aligned
            slabs = torch.full((S, M, ldc), float("nan"), device="cuda")
            _lib.check(lib.sca_gemm_kk_f32(Ad.data_ptr(), lda, Bd.data_ptr(), ldb, slabs.data_ptr(), ldc, M, N, K, S,
                                           M * ldc, st()), "gemm_kk")
            torch.cuda.synchronize()
            out = slabs[:, :, :N].double().sum(0).cpu()
            assert torch.isfinite(out).all(), (S, ldc)
            err = (out - ref).abs().max().item()
            assert err <= 2e-6 * scale, (S, ldc, err, scale)
            if ldc > N:                            # nothing written between the rows
                assert torch.isnan(slabs[:, :, N:]).all()


def test_gemm_kk_rejects_unsupported_operands(lib):
    """-2 for shapes / alignments the register-direct loads cannot take (the engine then keeps the tiled kernel)."""
    A = dev(torch.zeros(64, 136))
    Bm = dev(torch.zeros(64, 66))
    C = torch.empty(132, 64, device="cuda")
    ok = lambda *a: lib.sca_gemm_kk_supported(*a)
    assert ok(A.data_ptr(), 136, Bm.data_ptr(), 66, 132, 64) == 1
    assert ok(A.data_ptr(), 136, Bm.data_ptr(), 66, 130, 64) == 0          # M % 4 != 0
    assert ok(A.data_ptr(), 134, Bm.data_ptr(), 66, 132, 64) == 0          # lda % 4 != 0
    assert ok(A.data_ptr(), 136, Bm.data_ptr(), 65, 132, 64) == 0          # ldb % 2 != 0
    assert ok(A.data_ptr(), 136, Bm.data_ptr(), 66, 132, 63) == 0          # N % 2 != 0
    assert ok(A.data_ptr() + 4, 136, Bm.data_ptr(), 66, 132, 64) == 0      # A not 16-byte aligned
    assert ok(A.data_ptr(), 136, Bm.data_ptr() + 4, 66, 132, 64) == 0      # B not 8-byte aligned
    for args in ((A.data_ptr(), 136, Bm.data_ptr(), 66, C.data_ptr(), 64, 130, 64, 64, 1, 0),
                 (A.data_ptr() + 4, 136, Bm.data_ptr(), 66, C.data_ptr(), 64, 132, 64, 64, 1, 0),
                 (A.data_ptr(), 136, Bm.data_ptr() + 4, 66, C.data_ptr(), 64, 132, 64, 64, 1, 0)):
        assert lib.sca_gemm_kk_f32(*args, st()) == -2
    assert lib.sca_gemm_kk_f32(A.data_ptr(), 136, Bm.data_ptr(), 66, C.data_ptr(), 32, 132, 64, 64, 1, 0, st()) == -1


@pytest.mark.parametrize("S,n", [(128, 81920), (33, 1000), (64, 4 * 17), (40, 10240)])
def test_slab_reduce_many_slabs(lib, S, n):
    """sca_splitk_reduce with many slabs of a short vector (the MMD gradient slabs): the wide kernel (16 slab groups per
    workgroup, fixed summation order) against an fp64 sum, bit-identical from run to run."""
    from scarches_amd import _lib
    g = torch.Generator().manual_seed(S + n)
    slabs = dev(torch.randn(S, n, generator=g))
    out = torch.full((n,), float("nan"), device="cuda")
    _lib.check(lib.sca_splitk_reduce(slabs.data_ptr(), S, n, out.data_ptr(), n, st()), "reduce")
    out2 = torch.full((n,), float("nan"), device="cuda")
    _lib.check(lib.sca_splitk_reduce(slabs.data_ptr(), S, n, out2.data_ptr(), n, st()), "reduce")
    torch.cuda.synchronize()
    ref = slabs.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 1e-6 * slabs.abs().double().sum(0).max().item()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,N,K", [(6200, 2000, 128), (1000, 132, 64), (333, 2000, 100), (64, 128, 12)])
def test_gemm_short_k_row_major_epilogue(lib, M, N, K):
    """Forward GEMMs with a short contraction (the output layer): 64 x 64 tiles leave through LDS as 16-byte row-major
    stores when the output allows it (N % 4 == 0) - against fp64, with and without bias, ragged rows / columns."""
    from scarches_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    Ad, Wd, bd = dev(A), dev(W), dev(b)
    ref = A.double() @ W.double().t()
    scale = (A.abs().double() @ W.abs().double().t()).max().item()
    for bias in (None, bd):
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.sca_gemm_f32_cfg(0, 0, Ad.data_ptr(), K, Wd.data_ptr(), K, out.data_ptr(), N, M, N, K, 1, 0,
                                        bias.data_ptr() if bias is not None else None, 1, 1, st()), "gemm")
        torch.cuda.synchronize()
        r = ref + (b.double() if bias is not None else 0.0)
        err = (out.cpu().double() - r).abs().max().item()
        assert err <= 2e-6 * scale, (err, scale)


def test_gemm_bias_and_unaligned(lib):
    from scarches_amd import _lib
    M, N, K = 33, 50, 19           # lda/ldb not multiples of 4 -> scalar staging path
    A, W, b = torch.randn(M, K), torch.randn(N, K), torch.randn(N)
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.sca_gemm_f32(0, 0, dev(A).data_ptr(), K, dev(W).data_ptr(), K, out.data_ptr(), N, M, N, K, 1, 0,
                                dev(b).data_ptr(), st()), "gemm")
    ref = A.double() @ W.double().t() + b.double()
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5


def test_csr_gather(lib):
    from scarches_amd import _lib
    from oracle import cvae_oracle as O
    z = np.load(GOLDEN + "/host.npz")
    n, G = (int(v) for v in z["ds_shape"])
    rows = np.array([3, 10, 11, 40, 0, n - 1, 10], dtype=np.int64)
    x_ref, sf_ref = O.csr_rows_to_dense(z["ds_x_indptr"], z["ds_x_indices"], z["ds_x_data"], rows, G)
    for log1p in (0, 1):
        out = torch.full((len(rows), G), float("nan"), device="cuda")
        sf = torch.empty(len(rows), device="cuda")
        _lib.check(lib.sca_csr_gather_rows(dev(z["ds_x_indptr"]).data_ptr(), dev(z["ds_x_indices"]).data_ptr(),
                                           dev(z["ds_x_data"]).data_ptr(), dev(rows).data_ptr(), len(rows), G, log1p,
                                           out.data_ptr(), G, sf.data_ptr(), st()), "gather")
        torch.cuda.synchronize()
        ref = torch.log1p(x_ref) if log1p else x_ref
        if log1p:
            assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-6
        else:
            assert np.array_equal(out.cpu().numpy(), ref.numpy())       # raw counts: bit exact
        assert rel_err(sf.cpu().numpy(), sf_ref.numpy()) < 1e-6
    # the reference's own collate fixture (tests/golden/host.npz from AnnotatedDataset + custom_collate)
    rows = np.array([3, 10, 11, 40], dtype=np.int64)
    out = torch.empty(4, G, device="cuda")
    sf = torch.empty(4, device="cuda")
    _lib.check(lib.sca_csr_gather_rows(dev(z["ds_x_indptr"]).data_ptr(), dev(z["ds_x_indices"]).data_ptr(),
                                       dev(z["ds_x_data"]).data_ptr(), dev(rows).data_ptr(), 4, G, 0, out.data_ptr(), G,
                                       sf.data_ptr(), st()), "gather")
    assert np.array_equal(out.cpu().numpy(), z["ds_collate_x"])
    assert np.allclose(sf.cpu().numpy(), z["ds_collate_sizefactor"])


def test_csr_gather_empty_and_ragged(lib):
    from scarches_amd import _lib
    # rows with 0 nnz, one very long row, G not a multiple of 4
    G = 1001
    indptr = np.array([0, 0, 1001, 1001, 1003], dtype=np.int64)
    indices = np.concatenate([np.arange(1001), [5, 1000]]).astype(np.int32)
    data = np.concatenate([np.arange(1, 1002), [7, 9]]).astype(np.float32)
    out = torch.full((4, G), float("nan"), device="cuda")
    sf = torch.empty(4, device="cuda")
    _lib.check(lib.sca_csr_gather_rows(dev(indptr).data_ptr(), dev(indices).data_ptr(), dev(data).data_ptr(), None, 4,
                                       G, 0, out.data_ptr(), G, sf.data_ptr(), st()), "gather")
    o = out.cpu().numpy()
    assert (o[0] == 0).all() and (o[2] == 0).all()
    assert np.array_equal(o[1], np.arange(1, 1002, dtype=np.float32))
    assert o[3, 5] == 7 and o[3, 1000] == 9 and o[3].sum() == 16
    assert np.allclose(sf.cpu().numpy(), [0, data[:1001].sum(), 0, 16])


@pytest.mark.parametrize("key", list("abcde"))
def test_mmd_golden(lib, key):
    from scarches_amd import _lib
    from oracle import cvae_oracle as O
    z = np.load(GOLDEN + "/ops.npz")
    c = json.loads(str(z[f"mmd_{key}_cfg"]))
    y, lab = z[f"mmd_{key}_y"], z[f"mmd_{key}_c"]
    B, D, C = y.shape[0], y.shape[1], c["C"]
    ns, npart = lib.sca_mmd_num_slabs(B, D), lib.sca_mmd_num_partials(B, D)
    Wt = torch.empty(C * C, device="cuda")
    part = torch.zeros(npart, dtype=torch.float64, device="cuda")
    slabs = torch.full((ns, B, D), float("nan"), device="cuda")
    _lib.check(lib.sca_mmd_fwd_bwd(dev(y).data_ptr(), D, dev(lab, torch.int32).data_ptr(), B, D, C,
                                   -1 if c["boundary"] is None else c["boundary"], c["beta"], Wt.data_ptr(), None,
                                   part.data_ptr(), slabs.data_ptr(), B * D, st()), "mmd")
    torch.cuda.synchronize()
    val = part.sum().item()
    grad = slabs.sum(0).cpu().numpy()
    ref = float(z[f"mmd_{key}_val"])
    val64 = O.mmd_loss(torch.tensor(y).double(), torch.tensor(lab), C, c["beta"], c["boundary"]).item()
    # tolerance: rel 1e-5 of the reference value + the reference's own fp32 cancellation noise (vs the f64 oracle)
    assert abs(val - ref) <= 1e-5 * abs(ref) + 2 * abs(ref - val64) + 1e-6
    assert abs(val - val64) <= 1e-5 * abs(val64) + 1e-6
    if ref != 0:
        assert rel_err(grad, z[f"mmd_{key}_grad"]) < 5e-5


@pytest.fixture(params=[1, 2], ids=["rows-mfma", "symmetric-mfma"])
def mfma_form(lib, request):
    """The MFMA forms of the MMD kernels (1: row-block kernel, 2: symmetric whole-minibatch kernel; process-wide tuning
    hook), restored afterwards."""
    prev = lib.sca_mmd_select_form(request.param)
    yield
    torch.cuda.synchronize()
    lib.sca_mmd_select_form(prev)


@pytest.mark.parametrize("key", list("abcde"))
def test_mmd_golden_mfma_form(lib, key, mfma_form):
    """The same golden vectors through mmd_mfma_kernel (pair distances and owner gradients on v_mfma_f32_16x16x4_f32,
    near tiles - every tile of these small batches - on exact differences)."""
    test_mmd_golden(lib, key)


@pytest.mark.parametrize("B,D,C", [(1500, 10, 7), (4096, 10, 20), (777, 6, 3), (128, 10, 2), (200, 4, 5)])
def test_mmd_mfma_form_far_tiles_vs_fp64(lib, mfma_form, B, D, C):
    """Sizes at which most 16 x 16 tiles are far (d2 >= 2 everywhere: distances from the augmented-vector MFMA, owner
    gradients from the accumulator-as-operand MFMA) against a plain fp64 evaluation of the reference's formula
    (losses.py:166-270): value rel 1e-5, gradient 1e-5 of its maximum."""
    from scarches_amd import _lib
    g = torch.Generator().manual_seed(B + D)
    z = torch.randn(B, D, generator=g) * 1.3
    lab = torch.sort(torch.randint(0, C, (B,), generator=g))[0].to(torch.int32)
    ns, npart = lib.sca_mmd_num_slabs(B, D), lib.sca_mmd_num_partials(B, D)
    Wt = torch.empty(C * C, device="cuda")
    part = torch.zeros(npart, dtype=torch.float64, device="cuda")
    slabs = torch.full((ns, B, D), float("nan"), device="cuda")
    _lib.check(lib.sca_mmd_fwd_bwd(dev(z).data_ptr(), D, dev(lab, torch.int32).data_ptr(), B, D, C, -1, 1.0,
                                   Wt.data_ptr(), None, part.data_ptr(), slabs.data_ptr(), B * D, st()), "mmd")
    torch.cuda.synchronize()
    alphas = torch.tensor([1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1, 5, 10, 15, 20, 25, 30, 35, 100, 1e3, 1e4, 1e5, 1e6],
                          dtype=torch.float64, device="cuda")
    zz = z.double().cuda().requires_grad_(True)
    labd = lab.cuda()

    def K(x, y):
        d = ((x[:, None] - y[None]) ** 2).sum(2)
        return sum(torch.exp(-d / (2 * a)) for a in alphas)
    parts = [zz[labd == c] for c in range(C)]
    tot = 0
    for a in range(C):
        for b in range(a):
            if len(parts[a]) and len(parts[b]):
                tot = tot + K(parts[a], parts[a]).mean() + K(parts[b], parts[b]).mean() - 2 * K(parts[a], parts[b]).mean()
    tot.backward()
    val = part.sum().item()
    assert abs(val - tot.item()) <= 1e-5 * abs(tot.item()) + 1e-6
    got = slabs.double().sum(0)
    assert float((got - zz.grad).abs().max()) <= 1e-5 * float(zz.grad.abs().max())


@pytest.mark.parametrize("key", list("abcde"))
def test_mmd_row_blocks_golden(lib, key):
    """sca_mmd_rows_fwd_bwd (row block x all columns, the data-parallel form): ragged row blocks of the golden
    minibatches - incl. an empty one and padding rows with condition -1 between the blocks - add up to the reference's
    MMD value, and every block's gradient rows are the reference's gradient rows."""
    from scarches_amd import _lib
    from oracle import cvae_oracle as O
    z = np.load(GOLDEN + "/ops.npz")
    c = json.loads(str(z[f"mmd_{key}_cfg"]))
    y, lab = z[f"mmd_{key}_y"], z[f"mmd_{key}_c"]
    B, D, C = y.shape[0], y.shape[1], c["C"]
    cuts = [0, B // 3, B // 3, B - 1, B]                    # blocks of B/3, 0, ~2B/3 - 1 and 1 rows
    pad = 3                                                  # padding rows after every block
    ypad, lpad, starts = [], [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        starts.append(sum(len(a) for a in ypad))
        ypad += [y[lo:hi], np.full((pad, D), 7.5, dtype=np.float32)]
        lpad += [lab[lo:hi].astype(np.int32), np.full(pad, -1, dtype=np.int32)]
    yp, lp = dev(np.concatenate(ypad)), dev(np.concatenate(lpad), torch.int32)
    Bg = yp.shape[0]
    Wt = torch.empty(C * C, device="cuda")
    bnd = -1 if c["boundary"] is None else c["boundary"]
    val, ref_grad = 0.0, z[f"mmd_{key}_grad"]
    for (lo, hi), s0 in zip(zip(cuts[:-1], cuts[1:]), starts):
        n = hi - lo
        ns, npart = lib.sca_mmd_rows_num_slabs(Bg, n, D), lib.sca_mmd_rows_num_partials(Bg, n, D)
        if n == 0:
            assert ns == 0 and npart == 0
            assert lib.sca_mmd_rows_fwd_bwd(yp.data_ptr(), D, lp.data_ptr(), Bg, D, C, bnd, c["beta"], Wt.data_ptr(),
                                            None, s0, 0, None, None, 0, st()) == 0
            continue
        part = torch.full((npart,), float("nan"), dtype=torch.float64, device="cuda")
        slabs = torch.full((ns, n, D), float("nan"), device="cuda")
        _lib.check(lib.sca_mmd_rows_fwd_bwd(yp.data_ptr(), D, lp.data_ptr(), Bg, D, C, bnd, c["beta"], Wt.data_ptr(),
                                            None, s0, n, part.data_ptr(), slabs.data_ptr(), n * D, st()), "mmd_rows")
        torch.cuda.synchronize()
        val += part.sum().item()
        g = slabs.sum(0).cpu().numpy()
        if float(z[f"mmd_{key}_val"]) != 0:
            assert np.abs(g - ref_grad[lo:hi]).max() <= 5e-5 * np.abs(ref_grad).max(), (key, lo, hi)
    val64 = O.mmd_loss(torch.tensor(y).double(), torch.tensor(lab), C, c["beta"], c["boundary"]).item()
    assert abs(val - val64) <= 1e-5 * abs(val64) + 1e-6
    # argument checks: a row range outside the minibatch, a slab stride shorter than the block
    assert lib.sca_mmd_rows_fwd_bwd(yp.data_ptr(), D, lp.data_ptr(), Bg, D, C, bnd, 1.0, Wt.data_ptr(), None, Bg - 1, 2,
                                    None, None, 0, st()) == -1
    assert lib.sca_mmd_rows_fwd_bwd(yp.data_ptr(), D, lp.data_ptr(), Bg, D, C, bnd, 1.0, Wt.data_ptr(), None, 0, 2,
                                    Wt.data_ptr(), Wt.data_ptr(), D, st()) == -1


def test_nb_rows_against_oracle(lib):
    """softmax + NB rows kernel on random logits: per-row ll, dlogits, dtheta, dbias vs the f64 oracle."""
    from scarches_amd import _lib
    from oracle import cvae_oracle as O
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    # (B = 150 with 150 / 70 rows per workgroup: the kernel fetches row scalars 64 rows at a time and refills)
    for (B, G, C, zinb) in [(9, 37, 3, 0), (33, 300, 5, 0), (6, 2000, 4, 0), (17, 130, 3, 1), (5, 2500, 2, 0),
                            (150, 90, 4, 0), (150, 90, 4, 1)]:
        x = rng.poisson(rng.gamma(0.3, 3.0, size=(B, G))).astype(np.float32)
        x[0, :] = 0
        if G > 10:
            x[1, 3] = 3000.0
        xs = sp.csr_matrix(x)
        cond = np.sort(rng.integers(0, C, size=B)).astype(np.int32)
        logits = rng.normal(0, 1.5, size=(B, G)).astype(np.float32)
        pi = rng.normal(0, 2.0, size=(B, G)).astype(np.float32)
        theta = rng.normal(0, 1.2, size=(G, C)).astype(np.float32)
        sf = x.sum(1).astype(np.float32)
        # oracle (f64)
        lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
        pt = torch.tensor(pi, dtype=torch.float64, requires_grad=True)
        tt = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
        mu = torch.softmax(lt, -1) * torch.tensor(sf, dtype=torch.float64)[:, None]
        th = torch.exp(tt.T[torch.tensor(cond).long()])
        xt = torch.tensor(x, dtype=torch.float64)
        ll = O.zinb_loglik(xt, mu, th, pt) if zinb else O.nb_loglik(xt, mu, th)
        loss = -ll.sum(-1).mean()
        loss.backward()
        # HIP
        tabt = torch.empty(C, G, device="cuda")
        tabl = torch.empty(C, G, device="cuda")
        _lib.check(lib.sca_theta_prep(dev(theta).data_ptr(), G, C, tabt.data_ptr(), tabl.data_ptr(), st()), "prep")
        ld, pd_ = dev(logits), dev(pi)
        row_ll = torch.empty(B, device="cuda")
        dthT = torch.zeros(C, G, device="cuda")
        db = torch.zeros(G, device="cuda")
        dbp = torch.zeros(G, device="cuda")
        for rpb in (1, 4) if B < 100 else (1, 70, 150):
            ld.copy_(torch.tensor(logits)); pd_.copy_(torch.tensor(pi)); dthT.zero_(); db.zero_(); dbp.zero_()
            _lib.check(lib.sca_recon_rows(zinb, ld.data_ptr(), G, pd_.data_ptr() if zinb else None,
                                          dev(xs.indptr, torch.int64).data_ptr(), dev(xs.indices, torch.int32).data_ptr(),
                                          dev(xs.data).data_ptr(), None, dev(sf).data_ptr(), dev(cond).data_ptr(),
                                          tabt.data_ptr(), tabl.data_ptr(), B, G, C, rpb, row_ll.data_ptr(),
                                          dthT.data_ptr(), db.data_ptr(), dbp.data_ptr() if zinb else None, 1, None,
                                          None, st()),
                       "recon_rows")
            torch.cuda.synchronize()
            assert rel_err(row_ll.cpu().numpy(), ll.sum(-1).detach().numpy()) < 2e-6, (B, G, zinb)
            assert rel_err(ld.cpu().numpy(), lt.grad.numpy()) < 2e-5, (B, G, zinb)
            assert rel_err((dthT.t() * (-1.0 / B)).cpu().numpy(), tt.grad.numpy()) < 2e-5, (B, G, zinb)
            assert rel_err(db.cpu().numpy(), lt.grad.sum(0).numpy()) < 2e-5
            if zinb:
                assert rel_err(pd_.cpu().numpy(), pt.grad.numpy()) < 2e-5
                assert rel_err(dbp.cpu().numpy(), pt.grad.sum(0).numpy()) < 2e-5


def test_nb_elementwise_golden(lib):
    """The reference's own nb() values (tests/golden/ops.npz): feed mu directly via logits = log(mu), sf = sum(mu)."""
    from scarches_amd import _lib
    import scipy.sparse as sp
    z = np.load(GOLDEN + "/ops.npz")
    x, mu, theta = z["nb_x"], z["nb_mu"], z["nb_theta"]
    B, G = x.shape
    # theta varies per (b, g) in the fixture: give every row its own "condition"
    xs = sp.csr_matrix(x)
    logits = np.log(mu.astype(np.float64))
    sf = mu.astype(np.float64).sum(1)
    tabt = dev(theta)                                    # [C=B][G]
    tabl = dev(np.log(theta.astype(np.float64) + 1e-8).astype(np.float32))
    ld = dev(logits.astype(np.float32))
    row_ll = torch.empty(B, device="cuda")
    dthT = torch.zeros(B, G, device="cuda")
    _lib.check(lib.sca_recon_rows(0, ld.data_ptr(), G, None, dev(xs.indptr, torch.int64).data_ptr(),
                                  dev(xs.indices, torch.int32).data_ptr(), dev(xs.data).data_ptr(), None,
                                  dev(sf.astype(np.float32)).data_ptr(), dev(np.arange(B), torch.int32).data_ptr(),
                                  tabt.data_ptr(), tabl.data_ptr(), B, G, B, 1, row_ll.data_ptr(), dthT.data_ptr(),
                                  None, None, 1, None, None, st()), "recon_rows")
    torch.cuda.synchronize()
    ref = z["nb_ll"].astype(np.float64).sum(1)
    assert rel_err(row_ll.cpu().numpy(), ref) < 1e-5
    # theta * dll/dtheta
    assert rel_err(dthT.cpu().numpy(), theta * z["nb_dtheta"]) < 5e-5


def test_adam_kernel(lib):
    from scarches_amd import _lib
    from oracle import cvae_oracle as O
    n = 100003
    g0 = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g0)
    pd_, m, v = dev(p.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
    hyper = torch.zeros(8)
    for t in range(1, 4):
        g = torch.randn(n, generator=g0) * 0.1
        hyper[1], hyper[2], hyper[3], hyper[4], hyper[5] = 1e-3, 1 - 0.9 ** t, (1 - 0.999 ** t) ** 0.5, 0.01, 0.0
        hd = dev(hyper)
        _lib.check(lib.sca_adam_step(pd_.data_ptr(), dev(g).data_ptr(), m.data_ptr(), v.data_ptr(), n, hd.data_ptr(),
                                     0.9, 0.999, 0.04, st()), "adam")
        O.adam_step(pr, g, mr, vr, t, lr=1e-3, eps=0.01, wd=0.04)
    torch.cuda.synchronize()
    assert (pd_.cpu() - pr).abs().max().item() < 1e-6


@pytest.mark.parametrize("B,Bn", [(1000, 1000), (7, 300), (513, 0), (0, 5)])
def test_begin_step_batch_next_stages_scalars_and_both_minibatches(lib, B, Bn):
    """sca_begin_step_batch_next: hyper[0..5] <- the step's scalars, step_ctr += 1, rows / conditions of this minibatch
    and the row ids of the next one copied to their static buffers (bit-exact; B_next = 0 = sca_begin_step_batch)."""
    from scarches_amd import _lib
    g0 = torch.Generator().manual_seed(7)
    rows = dev(torch.randint(0, 1 << 40, (max(B, 1),), generator=g0, dtype=torch.int64))
    cond = dev(torch.randint(0, 100, (max(B, 1),), generator=g0, dtype=torch.int32))
    nxt = dev(torch.randint(0, 1 << 40, (max(Bn, 1),), generator=g0, dtype=torch.int64))
    rows_d, cond_d = dev(torch.full((max(B, 1) + 3,), -1, dtype=torch.int64)), dev(torch.full((max(B, 1) + 3,), -1, dtype=torch.int32))
    nxt_d = dev(torch.full((max(Bn, 1) + 3,), -1, dtype=torch.int64))
    hyper, ctr = dev(torch.zeros(8)), dev(torch.tensor([41], dtype=torch.int32))
    _lib.check(lib.sca_begin_step_batch_next(hyper.data_ptr(), ctr.data_ptr(), 0.25, 2e-3, 0.19, 0.0447, 0.01, 1.5,
                                             rows.data_ptr(), rows_d.data_ptr(), cond.data_ptr(), cond_d.data_ptr(), B,
                                             nxt.data_ptr() if Bn else None, nxt_d.data_ptr() if Bn else None, Bn, st()),
               "begin_step_batch_next")
    torch.cuda.synchronize()
    assert np.allclose(hyper.cpu().numpy()[:6], np.float32([0.25, 2e-3, 0.19, 0.0447, 0.01, 1.5]), rtol=0, atol=0)
    assert int(ctr.cpu()[0]) == 42
    assert torch.equal(rows_d[:B], rows[:B]) and bool((rows_d[B:] == -1).all())
    assert torch.equal(cond_d[:B], cond[:B]) and bool((cond_d[B:] == -1).all())
    assert torch.equal(nxt_d[:Bn], nxt[:Bn]) and bool((nxt_d[Bn:] == -1).all())
    # a next minibatch without buffers, negative sizes: rejected before any launch
    assert lib.sca_begin_step_batch_next(hyper.data_ptr(), ctr.data_ptr(), 0.25, 2e-3, 0.19, 0.0447, 0.01, 1.5,
                                         rows.data_ptr(), rows_d.data_ptr(), cond.data_ptr(), cond_d.data_ptr(), B,
                                         None, None, 4, st()) == -1
    assert lib.sca_begin_step_batch_next(hyper.data_ptr(), ctr.data_ptr(), 0.25, 2e-3, 0.19, 0.0447, 0.01, 1.5,
                                         rows.data_ptr(), rows_d.data_ptr(), cond.data_ptr(), cond_d.data_ptr(), -1,
                                         None, None, 0, st()) == -1


def test_mmd_reserve_lds_is_a_scheduling_knob_only(lib):
    """sca_mmd_reserve_lds: sizes outside [0, 65536] are rejected; the row-block MMD gives bit-identical losses and
    gradients whatever the reservation (it only caps the kernel's workgroups per CU)."""
    from scarches_amd import _lib
    assert lib.sca_mmd_reserve_lds(-1) == -1 and lib.sca_mmd_reserve_lds(65537) == -1
    B, D, C = 1500, 10, 6
    g0 = torch.Generator().manual_seed(3)
    z = dev(torch.randn(B, D, generator=g0))
    cond = dev(torch.sort(torch.randint(0, C, (B,), generator=g0)).values.to(torch.int32))
    outs = []
    try:
        for pad in (0, 24000, 40000, 65536):
            assert lib.sca_mmd_reserve_lds(pad) == 0
            Wt, cnt = dev(torch.zeros(C * C)), dev(torch.zeros(C, dtype=torch.int32))
            ns, npart = lib.sca_mmd_rows_num_slabs(B, B, D), lib.sca_mmd_rows_num_partials(B, B, D)
            part, slabs = dev(torch.zeros(npart, dtype=torch.float64)), dev(torch.zeros(ns * B * D))
            _lib.check(lib.sca_mmd_rows_fwd_bwd(z.data_ptr(), D, cond.data_ptr(), B, D, C, -1, 1.0, Wt.data_ptr(),
                                                cnt.data_ptr(), 0, B, part.data_ptr(), slabs.data_ptr(), B * D, st()),
                       "mmd_rows")
            torch.cuda.synchronize()
            outs.append((part.clone(), slabs.clone()))
    finally:
        lib.sca_mmd_reserve_lds(40000)
    for part, slabs in outs[1:]:
        assert torch.equal(part, outs[0][0]) and torch.equal(slabs, outs[0][1])


@pytest.mark.parametrize("Z", [10, 256, 280, 513])
def test_latent_fwd_narrow_and_wide_rows(Z):
    """sca_latent_fwd against the reference formulas in torch (models/base/_base.py:391-407 sampling,
    models/trvae/trvae.py:143-147 KL): the narrow kernel (Z <= 256) and the one-wave-per-row kernel that expiMap's wide
    latents (one per gene program, e.g. 280) take; two split-K slabs, bias, injected noise, ragged B."""
    from scarches_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B, S = 1031, 2
    gen = torch.Generator(device="cuda").manual_seed(Z)
    heads = torch.randn(S, B, 2 * Z, device="cuda", generator=gen) * 0.3
    bias = torch.randn(2 * Z, device="cuda", generator=gen) * 0.1
    eps = torch.randn(B, Z, device="cuda", generator=gen)
    mu_lv, eo, z, kl = (torch.empty(B, 2 * Z, device="cuda"), torch.empty(B, Z, device="cuda"),
                        torch.empty(B, Z, device="cuda"), torch.empty(B, device="cuda"))
    _lib.check(lib.sca_latent_fwd(heads.data_ptr(), S, B * 2 * Z, bias.data_ptr(), B, Z, eps.data_ptr(), 0, 0, None, 1,
                                  mu_lv.data_ptr(), eo.data_ptr(), z.data_ptr(), kl.data_ptr(), st), "latent_fwd")
    torch.cuda.synchronize()
    h = (heads.double().sum(0) + bias.double())
    mu, lv = h[:, :Z], h[:, Z:]
    var = lv.exp() + 1e-4
    z_ref = mu + eps.double() * var.sqrt()
    kl_ref = (0.5 * (var + mu * mu - 1 - var.log())).sum(1)
    assert float((mu_lv.double() - h).abs().max()) <= 1e-6
    assert torch.equal(eo, eps)
    assert float((z.double() - z_ref).abs().max()) <= 1e-5
    assert float(((kl.double() - kl_ref).abs() / kl_ref.abs().clamp_min(1e-3)).max()) <= 1e-5


def test_theta_prep_zero_matches_theta_prep_and_fills():
    """sca_theta_prep_zero = sca_theta_prep + two zero fills in one launch: identical tables, both ranges zeroed and
    nothing beyond them touched; NULL ranges, G * C == 0 and misaligned ranges (-1)."""
    from scarches_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    G, C = 333, 7
    theta = torch.randn(G, C, device="cuda")
    t0, l0 = torch.empty(C, G, device="cuda"), torch.empty(C, G, device="cuda")
    t1, l1 = torch.full((C, G), float("nan"), device="cuda"), torch.full((C, G), float("nan"), device="cuda")
    _lib.check(lib.sca_theta_prep(theta.data_ptr(), G, C, t0.data_ptr(), l0.data_ptr(), st), "theta_prep")
    za = torch.full((4 * 1000 + 8,), 3.0, device="cuda")
    zb = torch.full((4 * 70001 + 8,), 5.0, device="cuda")
    _lib.check(lib.sca_theta_prep_zero(theta.data_ptr(), G, C, t1.data_ptr(), l1.data_ptr(), za.data_ptr(), 16 * 1000,
                                       zb.data_ptr(), 16 * 70001, st), "theta_prep_zero")
    torch.cuda.synchronize()
    assert torch.equal(t0, t1) and torch.equal(l0, l1)
    assert float(za[:4000].abs().max()) == 0.0 and torch.all(za[4000:] == 3.0)
    assert float(zb[:4 * 70001].abs().max()) == 0.0 and torch.all(zb[4 * 70001:] == 5.0)
    # tables only / fills only
    t1.fill_(float("nan"))
    _lib.check(lib.sca_theta_prep_zero(theta.data_ptr(), G, C, t1.data_ptr(), l1.data_ptr(), None, 0, None, 0, st), "tpz")
    za.fill_(3.0)
    _lib.check(lib.sca_theta_prep_zero(None, 0, 0, None, None, za.data_ptr(), 16 * 10, None, 0, st), "tpz")
    torch.cuda.synchronize()
    assert torch.equal(t0, t1) and float(za[:40].abs().max()) == 0.0 and torch.all(za[40:] == 3.0)
    assert lib.sca_theta_prep_zero(theta.data_ptr(), G, C, t1.data_ptr(), l1.data_ptr(), za.data_ptr() + 4, 16, None, 0, st) == -1
    assert lib.sca_theta_prep_zero(theta.data_ptr(), G, C, t1.data_ptr(), l1.data_ptr(), za.data_ptr(), 20, None, 0, st) == -1
