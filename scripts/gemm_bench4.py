"""Dev probe: tile / split-K sweep of the five wide GEMM shapes of a cfg2 step at B cells (main library)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [('enc_fwd', 0, 0, B, 128, 2000), ('out_fwd', 0, 0, B, 2000, 128), ('out_dX', 0, 1, B, 128, 2000),
          ('out_dW', 1, 1, 2000, 128, B), ('enc_dW', 1, 1, 128, 2000, B)]
fn = lib.sca_gemm_f32_cfg
for name, am, bm, M, N, K in shapes:
    A = torch.randn(M * K, device='cuda'); Bm = torch.randn(N * K, device='cuda')
    lda = K if am == 0 else M; ldb = K if bm == 0 else N
    res = []
    for wm, wn in ((2, 2), (1, 2), (2, 1), (1, 1)):
        for S in (1, 2, 4, 8, 16, 32):
            if S > 1 and (K // 32) // S < 2: continue
            C = torch.empty(S * M * N, device='cuda')
            for _ in range(2):
                fn(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10 * 1e3, wm, wn, S))
    res.sort()
    fl = 2.0 * M * N * K
    heur = lib.sca_gemm_splitk_heuristic(M, N, K)
    print(f"B={B} {name:8s} heurS={heur} best: " + "  ".join(f"({wm}x{wn},S={S}) {t:.1f}us {fl/t/1e6:.0f}TF" for t, wm, wn, S in res[:6]), flush=True)
    print("      1x1: " + "  ".join(f"S={S} {t:.1f}" for t, wm, wn, S in sorted(res, key=lambda r: r[3]) if (wm, wn) == (1, 1)), flush=True)
