"""Isolated timings of the tile-skipping masked decoder kernels (csrc/masked_gemm.hip) against the dense fp32 GEMMs
they replace, at the expiMap cfg3 shape (5000 genes x 280 gene programs, density 1.2 %).

    python scripts/masked_bench.py [B] [G] [L]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib                                # noqa: E402


def plan(pat):
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
    return NT, Kmax, torch.tensor(k.astype(np.int32)).cuda(), torch.tensor(cols).cuda(), float(k.sum()) / (NT * Zt)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    Zt = int(sys.argv[3]) if len(sys.argv) > 3 else 280
    L = _lib.load()
    rng = np.random.default_rng(0)
    sizes = np.clip(rng.lognormal(np.log(30), 0.8, Zt), 13, 300).astype(int)
    pat = np.zeros((G, Zt), bool)
    for k, s in enumerate(sizes):
        pat[rng.choice(G, s, replace=False), k] = True
    pat[np.arange(G), rng.integers(0, Zt, G)] = True
    NT, Kmax, tk, tc, frac = plan(pat)
    meta = (tk.data_ptr(), tc.data_ptr(), NT, Kmax)
    W = torch.randn(G, Zt, device="cuda") * torch.tensor(pat).cuda()
    z = torch.randn(B, Zt, device="cuda")
    dl = torch.randn(B, G, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    n = NT * Kmax * 32
    Wc = torch.empty(2 * n, device="cuda")
    out = torch.empty(B, G, device="cuda")
    Sx, Sw = L.sca_masked_dx_slabs(B, NT), L.sca_masked_dw_slabs(B, NT)
    dzs = torch.empty(Sx, B, Zt, device="cuda")
    dWc = torch.empty(Sw, n, device="cuda")
    dWeff = torch.zeros(G, Zt, device="cuda")
    L.sca_masked_pack(W.data_ptr(), Zt, *meta, G, Zt, Wc.data_ptr(), st)
    print(f"B {B}  G {G}  L {Zt}  density {pat.mean():.4f}  NT {NT}  Kmax {Kmax}  active k-steps {frac:.3f}  "
          f"dx slabs {Sx}  dw slabs {Sw}")
    flop = 2.0 * B * G * Zt
    res = {
        "pack": timeit(lambda: L.sca_masked_pack(W.data_ptr(), Zt, *meta, G, Zt, Wc.data_ptr(), st)),
        "fwd": timeit(lambda: L.sca_masked_fwd(z.data_ptr(), Zt, Wc.data_ptr(), *meta, B, G, Zt, out.data_ptr(), G, st)),
        "dx": timeit(lambda: L.sca_masked_dx(dl.data_ptr(), G, Wc.data_ptr(), *meta, B, G, Zt, dzs.data_ptr(), B * Zt,
                                             st)),
        "dw": timeit(lambda: L.sca_masked_dw(z.data_ptr(), Zt, dl.data_ptr(), G, *meta, B, G, Zt, dWc.data_ptr(), n,
                                             st)),
        "unpack": timeit(lambda: L.sca_masked_unpack(dWc.data_ptr(), Sw, n, *meta, G, Zt, dWeff.data_ptr(), Zt, st)),
    }
    # the dense products through the library's own GEMM
    slab = torch.empty(64 * max(B * G, G * Zt, B * Zt) // 8, device="cuda")

    def dense(am, bm, A, lda, Bm, ldb, M, N, K):
        S = L.sca_gemm_splitk_heuristic(M, N, K)
        assert S * M * N <= slab.numel()
        return lambda: L.sca_gemm_f32(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, slab.data_ptr(), N, M, N, K, S,
                                      M * N, None, st)
    res["dense fwd"] = timeit(dense(0, 0, z, Zt, W, Zt, B, G, Zt))
    res["dense dx"] = timeit(dense(0, 1, dl, G, W, Zt, B, Zt, G))
    res["dense dw"] = timeit(dense(1, 1, dl, G, z, Zt, G, Zt, B))
    for k, v in res.items():
        extra = ""
        if k in ("fwd", "dx", "dw"):
            extra = f"   {flop * frac / v / 1e6:7.1f} TFLOP/s issued   {flop / v / 1e6:7.1f} TFLOP/s dense-equivalent"
        elif k.startswith("dense"):
            extra = f"   {flop / v / 1e6:7.1f} TFLOP/s"
        print(f"{k:10s} {v:8.1f} us{extra}")


if __name__ == "__main__":
    main()
