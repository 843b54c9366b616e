import sys, ctypes, torch
sys.path.insert(0, '.')
from scarches_amd import _lib
from scarches_amd.synth import make_synthetic_csr
st = torch.cuda.current_stream().cuda_stream
N, G, C, B = 100000, 2000, 20, 4096
indptr, indices, data, cond, _ = make_synthetic_csr(N, G, C, 'cuda', seed=0)
ids = torch.randint(0, N, (B,), device='cuda'); c = cond[ids]; o = torch.argsort(c, stable=True); ids, c = ids[o].contiguous(), c[o].contiguous()
logits0 = torch.randn(B, G, device='cuda'); sf = torch.rand(B, device='cuda') * 3000 + 500
tabt = torch.exp(torch.randn(C, G, device='cuda')); tabl = torch.log(tabt + 1e-8)
row_ll = torch.empty(B, device='cuda'); dth = torch.zeros(C, G, device='cuda'); db = torch.zeros(G, device='cuda')
for variant in sys.argv[1:]:
    lib = ctypes.CDLL(f'scarches_amd/csrc/libsca_{variant}.so')
    fn = lib.sca_recon_rows; fn.argtypes = _lib.SIGNATURES['sca_recon_rows']
    for rpb in (1, 2, 4, 8):
        logits = logits0.clone()
        for _ in range(2):
            fn(0, logits.data_ptr(), G, None, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), ids.data_ptr(), sf.data_ptr(), c.data_ptr(), tabt.data_ptr(), tabl.data_ptr(), B, G, C, rpb, row_ll.data_ptr(), dth.data_ptr(), db.data_ptr(), None, 1, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(0, logits.data_ptr(), G, None, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), ids.data_ptr(), sf.data_ptr(), c.data_ptr(), tabt.data_ptr(), tabl.data_ptr(), B, G, C, rpb, row_ll.data_ptr(), dth.data_ptr(), db.data_ptr(), None, 1, st)
        e1.record(); torch.cuda.synchronize()
        print(f"{variant:14s} rpb={rpb}: {e0.elapsed_time(e1)/10*1e3:.1f} us")
