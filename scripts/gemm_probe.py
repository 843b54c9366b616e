"""Dev probe: time the five big GEMM shapes of the cfg2 step over tile / split-K configurations (HIP events)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scarches_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2:      # a GEMM-only variant library (scripts/_variants/*.so)
    import ctypes
    V = ctypes.CDLL(os.path.abspath(sys.argv[2]))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    V.sca_gemm_f32_cfg.argtypes = [i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i64, vp, i32, i32, vp]
    V.sca_gemm_splitk_heuristic.argtypes = [i32, i32, i32]
    L = V
    print("variant", sys.argv[2])
G, H = 2000, 128
# (name, amode, bmode, M, N, K, lda, ldb)
shapes = [
    ("enc0 fwd  x W^T", 0, 0, B, H, G, G, G),
    ("out  fwd  h W^T", 0, 0, B, G, H, H, H),
    ("out  dX   dY W ", 0, 1, B, H, G, G, H),
    ("out  dW   dY^T h", 1, 1, G, H, B, G, H),
    ("enc0 dW   da^T x", 1, 1, H, G, B, H, G),
    ("mid  dW   da^T h", 1, 1, H, H, B, H, H),
]
A = torch.randn(B * G, device=dev)
Bm = torch.randn(B * G, device=dev)
C = torch.empty(64 * 1024 * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
for name, am, bm, M, N, K, lda, ldb in shapes:
    flop = 2.0 * M * N * K
    res = []
    Sh = L.sca_gemm_splitk_heuristic(M, N, K)
    quick = os.environ.get("GEMM_QUICK") == "1"
    for wm, wn, S in ([(0, 0, Sh)] if quick else itertools.product((1, 2), (1, 2), (1, 2, 4, 8, 16, 32))):
        if S * M * N > C.numel() or K // S < 32:
            continue
        args = (am, bm, A.data_ptr(), lda, Bm.data_ptr(), ldb, C.data_ptr(), N, M, N, K, S, M * N, None, wm, wn, st)
        for _ in range(3):
            L.sca_gemm_f32_cfg(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.sca_gemm_f32_cfg(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append((us, wm, wn, S))
    res.sort()
    best = ", ".join(f"{64*wm}x{64*wn} S={S}: {us:.1f}us ({flop/us/1e6:.0f} TF)" for us, wm, wn, S in res[:4])
    print(f"{name} M={M} N={N} K={K} heuristic S={Sh} | {best}", flush=True)
