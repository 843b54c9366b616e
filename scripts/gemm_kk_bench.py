"""Dev probe: the register-direct k-major GEMM (sca_gemm_kk_f32) against the LDS-staged kernel on the weight-gradient
shapes of a cfg2 step, with a correctness check against torch (fp64)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [('out_dW', 2000, 128, B), ('enc_dW', 128, 2000, B), ('mid_dW', 128, 128, B), ('heads_dW', 20, 128, B),
          ('odd', 332, 78, 1001), ('odd2', 1000, 130, 5000)]


def timeit(f, n=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K in shapes:
    A = torch.randn(K, M, device='cuda'); Bm = torch.randn(K, N, device='cuda')
    ref = (A.double().t() @ Bm.double())
    fl = 2.0 * M * N * K
    S0 = lib.sca_gemm_splitk_heuristic(M, N, K)
    C0 = torch.empty(S0 * M * N, device='cuda')
    t0 = timeit(lambda: lib.sca_gemm_f32(1, 1, A.data_ptr(), M, Bm.data_ptr(), N, C0.data_ptr(), N, M, N, K, S0, M * N, None, st))
    out = []
    Sh = lib.sca_gemm_kk_splitk_heuristic(M, N, K)
    for S in sorted({1, 2, 4, 6, 8, 12, 16, 24, 32, Sh}):
        if S > 1 and K // (4 * S) < 48:
            continue
        C = torch.full((S * M * N,), float('nan'), device='cuda')
        rc = lib.sca_gemm_kk_f32(A.data_ptr(), M, Bm.data_ptr(), N, C.data_ptr(), N, M, N, K, S, M * N, st)
        assert rc == 0, rc
        got = C.view(S, M, N).double().sum(0)
        err = float((got - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: lib.sca_gemm_kk_f32(A.data_ptr(), M, Bm.data_ptr(), N, C.data_ptr(), N, M, N, K, S, M * N, st))
        out.append((S, t, err))
    print(f"{name:9s} M={M} N={N} K={K}  lds-kernel S={S0}: {t0:.1f}us {fl/t0/1e6:.0f}TF | kk (heur S={Sh}): " +
          "  ".join(f"S={S}:{t:.1f}us/{fl/t/1e6:.0f}TF/err{err:.0e}" for S, t, err in out), flush=True)
