import sys, torch
sys.path.insert(0, '.')
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [('enc_fwd', 0, 0, B, 128, 2000), ('out_fwd', 0, 0, B, 2000, 128), ('out_dX', 0, 1, B, 128, 2000),
          ('enc_dW', 1, 1, 128, 2000, B), ('out_dW', 1, 1, 2000, 128, B), ('hid_fwd', 0, 0, B, 128, 128),
          ('hid_dW', 1, 1, 128, 128, B), ('hid_dX', 0, 1, B, 128, 128), ('heads_fwd', 0, 0, B, 20, 128),
          ('dec0_fwd', 0, 0, B, 128, 10), ('heads_dW', 1, 1, 20, 128, B)]
for name, am, bm, M, N, K in shapes:
    A = torch.randn(M * K, device='cuda'); Bm = torch.randn(N * K, device='cuda')
    lda = K if am == 0 else M; ldb = K if bm == 0 else N
    res = []
    for wm, wn in ((2, 2), (1, 2), (2, 1), (1, 1)):
        for S in (1, 2, 4, 8, 16, 32, 64):
            if S > 1 and ((K + 15) // 16) // S < 2: continue
            C = torch.empty(S * M * N, device='cuda')
            for _ in range(2):
                lib.sca_gemm_f32_cfg(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.sca_gemm_f32_cfg(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10 * 1e3
            res.append((t, wm, wn, S))
    res.sort()
    hS = lib.sca_gemm_splitk_heuristic(M, N, K)
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M} N={N} K={K}  heurS={hS}  best: " + "  ".join(f"({wm}x{wn},S={S}) {t:.1f}us {fl/t/1e6:.1f}TF" for t, wm, wn, S in res[:5]))
