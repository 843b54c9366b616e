"""Dev probe: throughput of the kNN label transfer (top-50 in Z=10) for reference-atlas sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scarches_amd.utils.knn import HipKNeighbors
from scarches_amd import _lib

lib = _lib.load()
for nr, nq in ((100_000, 10_000), (1_000_000, 65_536), (2_300_000, 65_536)):
    ref = torch.randn(nr, 10, device="cuda")
    qry = torch.randn(nq, 10, device="cuda")
    m = HipKNeighbors(ref.cpu().numpy(), 50)
    lab = torch.randint(0, 30, (nr,), device="cuda", dtype=torch.int32)
    pred = torch.empty(nq, dtype=torch.int32, device="cuda"); prob = torch.empty(nq, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    m.kneighbors_device(qry); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    dist, idx = m.kneighbors_device(qry)
    e[1].record()
    lib.sca_knn_vote(dist.data_ptr(), idx.data_ptr(), nq, 50, lab.data_ptr(), pred.data_ptr(), prob.data_ptr(), st)
    e[2].record(); torch.cuda.synchronize()
    t1, t2 = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    print(f"Nr={nr} Nq={nq}: topk {t1:.2f} ms ({3*10*nr*nq/t1/1e9:.1f} TFLOP/s of 3*D flop/pair, {nq/t1*1e3/1e6:.2f} M queries/s), vote {t2:.2f} ms", flush=True)
