import sys, ctypes, torch
sys.path.insert(0, '.')
from scarches_amd import _lib
st = torch.cuda.current_stream().cuda_stream
B = 4096
shapes = [('enc_fwd', 0, 0, B, 128, 2000), ('out_fwd', 0, 0, B, 2000, 128), ('out_dX', 0, 1, B, 128, 2000),
          ('enc_dW', 1, 1, 128, 2000, B), ('hid_fwd', 0, 0, B, 128, 128), ('cube', 0, 0, 4096, 4096, 4096)]
for variant in sys.argv[1:]:
    lib = ctypes.CDLL(f'scarches_amd/csrc/libsca_{variant}.so')
    fn = lib.sca_gemm_f32_cfg
    fn.argtypes = _lib.SIGNATURES['sca_gemm_f32_cfg']
    for name, am, bm, M, N, K in shapes:
        A = torch.randn(M * K, device='cuda'); Bm = torch.randn(N * K, device='cuda')
        lda = K if am == 0 else M; ldb = K if bm == 0 else N
        res = []
        for wm, wn in ((2, 2), (1, 2), (1, 1)):
            for S in (1, 2, 4, 8, 16, 32):
                if S > 1 and (K // 32) // S < 2: continue
                if name == 'cube' and S > 1: continue
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
        print(f"{variant:10s} {name:8s} best: " + "  ".join(f"({wm}x{wn},S={S}) {t:.1f}us {fl/t/1e6:.0f}TF" for t, wm, wn, S in res[:3]))
