"""Dev probe: time the CSR gather kernel alone (SCA_GATHER_T selects the threads per cell)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
from scarches_amd.synth import make_synthetic_csr
lib = _lib.load()
dev = torch.device("cuda", 0)
G = 2000
indptr, indices, data, cond_all, _ = make_synthetic_csr(200_000, G, 20, dev, seed=1000)
st = torch.cuda.current_stream().cuda_stream
for B in (4096, 8192, 16384):
    ids = torch.randint(0, 200_000, (B,), device=dev, dtype=torch.int64)
    x = torch.empty(B, G, device=dev); sf = torch.empty(B, device=dev)
    def run():
        lib.sca_csr_gather_rows(indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), ids.data_ptr(), B, G, 1,
                                x.data_ptr(), G, sf.data_ptr(), st)
    run(); run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nnz = (indptr[ids + 1] - indptr[ids]).sum().item()
    byt = 8 * nnz + 24 * B + 4 * G * B + 4 * B
    ref = torch.log1p(torch.sparse_csr_tensor(indptr, indices, data, size=(200_000, G)).to_dense()[ids]) if B == 4096 else None
    ok = "" if ref is None else f" bitexact={torch.equal(ref, x)}"
    print(f"T={os.environ.get('SCA_GATHER_T','auto')} B={B}: {us:.1f} us  {byt/us/1e6:.2f} TB/s{ok}", flush=True)
