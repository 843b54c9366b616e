"""sca_knn_topk_static (the entry point that allocates nothing, include/scarches_hip.h) against sca_knn_topk through
the C ABI: bit-identical distances and indices on two calls in a row, the second with a smaller input (the static
workspace keeps the first call's contents), and a clear error beyond the static workspace's limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _topk_pair(lib, _lib, ref, qry, k):
    nr, d = ref.shape
    nq = qry.shape[0]
    chunks = lib.sca_knn_num_chunks(nr, nq)
    pd2 = torch.empty(chunks * nq * k, device="cuda")
    pidx = torch.empty(chunks * nq * k, dtype=torch.int32, device="cuda")
    out = []
    for static in (False, True):
        dist = torch.full((nq, k), -1.0, device="cuda")
        idx = torch.full((nq, k), -1, dtype=torch.int32, device="cuda")
        if static:
            rc = lib.sca_knn_topk_static(ref.data_ptr(), nr, qry.data_ptr(), nq, d, k, dist.data_ptr(),
                                         idx.data_ptr(), None)
        else:
            rc = lib.sca_knn_topk(ref.data_ptr(), nr, qry.data_ptr(), nq, d, k, pd2.data_ptr(), pidx.data_ptr(),
                                  dist.data_ptr(), idx.data_ptr(), None)
        _lib.check(rc, "knn_topk")
        torch.cuda.synchronize()
        out.append((dist, idx))
    return out


def test_knn_topk_static_matches_knn_topk():
    from scarches_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(11)
    # two calls in a row; the second is smaller in every dimension (fewer chunks, fewer queries, smaller k)
    for nr, nq, d, k in ((60_000, 3_000, 10, 50), (9_000, 700, 10, 15)):
        ref = torch.randn(nr, d, device="cuda", generator=gen)
        qry = torch.randn(nq, d, device="cuda", generator=gen)
        assert lib.sca_knn_num_chunks(nr, nq) * nq * k <= lib.sca_knn_static_max_entries()
        (d0, i0), (d1, i1) = _topk_pair(lib, _lib, ref, qry, k)
        assert torch.equal(d0, d1) and torch.equal(i0, i1), (nr, nq, k)
        assert float(d1.min()) >= 0.0 and int(i1.min()) >= 0 and int(i1.max()) < nr
        # sanity against torch: the k-th distance of a few queries
        dd = torch.cdist(qry[:64].double(), ref.double()).topk(k, largest=False).values.float()
        assert torch.allclose(d1[:64], dd, rtol=1e-5, atol=1e-6)


def test_knn_topk_static_rejects_sizes_beyond_its_workspace():
    from scarches_amd import _lib
    lib = _lib.load()
    nr, nq, d, k = 4096, 70_000, 10, 50         # 1 chunk x 70 000 queries x 50 > 2^21 entries
    assert lib.sca_knn_num_chunks(nr, nq) * nq * k > lib.sca_knn_static_max_entries()
    ref = torch.randn(nr, d, device="cuda")
    qry = torch.randn(nq, d, device="cuda")
    dist = torch.empty(nq, k, device="cuda")
    idx = torch.empty(nq, k, dtype=torch.int32, device="cuda")
    rc = lib.sca_knn_topk_static(ref.data_ptr(), nr, qry.data_ptr(), nq, d, k, dist.data_ptr(), idx.data_ptr(), None)
    assert rc == -3
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(rc, "knn_topk_static")


def test_knn_topk_static_inside_a_graph_capture_after_the_warm_up_call():
    """The Nq = 0 warm-up call does the two steps that are not stream operations (address lookup, kernel attributes);
    after it the entry point only launches kernels and can sit inside a HIP graph: replaying the graph on new inputs
    written into the captured buffers gives sca_knn_topk's result bit for bit."""
    from scarches_amd import _lib
    lib = _lib.load()
    assert lib.sca_knn_topk_static(None, 0, None, 0, 0, 0, None, None, None) == 0       # warm-up, outside capture
    nr, nq, d, k = 20_000, 1_000, 10, 20
    gen = torch.Generator(device="cuda").manual_seed(5)
    ref = torch.randn(nr, d, device="cuda", generator=gen)
    qry = torch.randn(nq, d, device="cuda", generator=gen)
    dist = torch.empty(nq, k, device="cuda")
    idx = torch.empty(nq, k, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = lib.sca_knn_topk_static(ref.data_ptr(), nr, qry.data_ptr(), nq, d, k, dist.data_ptr(), idx.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    qry.copy_(torch.randn(nq, d, device="cuda", generator=gen))     # new queries into the captured buffer
    g.replay()
    torch.cuda.synchronize()
    chunks = lib.sca_knn_num_chunks(nr, nq)
    pd2 = torch.empty(chunks * nq * k, device="cuda")
    pidx = torch.empty(chunks * nq * k, dtype=torch.int32, device="cuda")
    d0 = torch.empty(nq, k, device="cuda")
    i0 = torch.empty(nq, k, dtype=torch.int32, device="cuda")
    _lib.check(lib.sca_knn_topk(ref.data_ptr(), nr, qry.data_ptr(), nq, d, k, pd2.data_ptr(), pidx.data_ptr(),
                                d0.data_ptr(), i0.data_ptr(), None), "knn_topk")
    torch.cuda.synchronize()
    assert torch.equal(d0, dist) and torch.equal(i0, idx)
