"""Dev probe: the five wide GEMMs of a cfg2 step at the planner's own split (for rocprofv3 --pmc passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [('enc_fwd', 0, 0, B, 128, 2000), ('out_fwd', 0, 0, B, 2000, 128), ('out_dX', 0, 1, B, 128, 2000),
          ('out_dW', 1, 1, 2000, 128, B), ('enc_dW', 1, 1, 128, 2000, B)]
for name, am, bm, M, N, K in shapes:
    A = torch.randn(M * K, device='cuda'); Bm = torch.randn(N * K, device='cuda')
    lda = K if am == 0 else M; ldb = K if bm == 0 else N
    S = lib.sca_gemm_splitk_heuristic(M, N, K)
    C = torch.empty(S * M * N, device='cuda')
    for _ in range(5):
        lib.sca_gemm_f32(am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, st)
    torch.cuda.synchronize()
    print(name, S, flush=True)
