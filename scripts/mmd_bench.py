import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
C, D = 20, 10
for B in (4096, 8192, 16384):
    for scale in (1.0,):
        z = torch.randn(B, D, device='cuda') * scale
        cond = torch.sort(torch.randint(0, C, (B,), device='cuda', dtype=torch.int32))[0].contiguous()
        ns, npart = lib.sca_mmd_num_slabs(B, D), lib.sca_mmd_num_partials(B, D)
        Wt = torch.empty(C * C, device='cuda'); part = torch.zeros(npart, dtype=torch.float64, device='cuda')
        slabs = torch.empty(ns * B * D, device='cuda')
        for g in (slabs.data_ptr(), None):
            def run():
                lib.sca_mmd_fwd_bwd(z.data_ptr(), D, cond.data_ptr(), B, D, C, -1, 1.0, Wt.data_ptr(), None, part.data_ptr(), g, B * D, st)
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} scale={scale} grad={g is not None}: {e0.elapsed_time(e1)/10*1e3:.1f} us  ns={ns}")
