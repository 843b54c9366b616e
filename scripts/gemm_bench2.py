import sys, torch
sys.path.insert(0, '.')
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def run(name, am, bm, M, N, K, wm, wn, S, zero=False):
    A = (torch.zeros if zero else torch.randn)(M * K, device='cuda'); Bm = (torch.zeros if zero else torch.randn)(N * K, device='cuda')
    lda = K if am == 0 else M; ldb = K if bm == 0 else N
    C = torch.empty(S * M * N, device='cuda')
    for _ in range(2):
        lib.sca_gemm_f32_cfg(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.sca_gemm_f32_cfg(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{name:12s} mode({am},{bm}) M={M} N={N} K={K} tile({wm}x{wn}) S={S} zero={zero}: {t:.1f}us {2.0*M*N*K/t/1e6:.1f} TF")
for am, bm in ((0, 0), (1, 1), (0, 1)):
    run('cube4096', am, bm, 4096, 4096, 4096, 2, 2, 1)
run('cube4096', 0, 0, 4096, 4096, 4096, 1, 1, 1)
run('cube4096z', 0, 0, 4096, 4096, 4096, 2, 2, 1, zero=True)
run('cube2048', 0, 0, 2048, 2048, 2048, 2, 2, 1)
run('skinnyK', 0, 0, 4096, 128, 16000, 2, 2, 16)
run('enc_fwd_z', 0, 0, 4096, 128, 2000, 2, 2, 16, zero=True)
run('enc_fwd', 0, 0, 4096, 128, 2000, 2, 2, 16)
run('enc_fwd16k', 0, 0, 16384, 128, 2000, 2, 2, 4)
run('enc_fwd16k', 0, 0, 16384, 128, 2000, 1, 1, 2)
