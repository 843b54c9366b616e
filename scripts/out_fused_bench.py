"""Dev probe: sca_out_nb_fused alone against sca_gemm_f32 + sca_recon_rows (cfg2 shapes, synthetic counts)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scarches_amd import _lib
lib = _lib.load()
B, G, K, C = int(os.environ.get("B", 8192)), 2000, 128, 20
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(B, K, device="cuda", generator=g).relu_() * 0.8
W = torch.randn(G, K, device="cuda", generator=g) * 0.1
bias = torch.randn(G, device="cuda", generator=g) * 0.3
theta = torch.randn(G, C, device="cuda", generator=g)
cond = torch.randint(0, C, (B,), device="cuda", generator=g).to(torch.int32).sort().values
mask = torch.rand(B, G, device="cuda", generator=g) < 0.145
x = (torch.poisson(torch.rand(B, G, device="cuda", generator=g) * 12.0 * torch.rand(B, 1, device="cuda", generator=g) * 2) + 1.0) * mask
csr = x.to_sparse_csr()
ip, ix, dv = csr.crow_indices(), csr.col_indices().to(torch.int32), csr.values()
sf = x.sum(1)
print("nnz/row", dv.numel() / B, "frac > 16:", float((dv > 16).float().mean()))
tab_theta, tab_lte = torch.empty(C, G, device="cuda"), torch.empty(C, G, device="cuda")
lib.sca_theta_prep(theta.data_ptr(), G, C, tab_theta.data_ptr(), tab_lte.data_ptr(), None)
gam = torch.empty(C, G, 17, 2, device="cuda")
logits = torch.empty(B, G, device="cuda"); ovf = torch.empty(B, G, device="cuda")
row_ll = torch.empty(B, device="cuda"); dthetaT = torch.zeros(C, G, device="cuda"); dbias = torch.zeros(G, device="cuda")
def unfused():
    lib.sca_gemm_f32(0, 0, h.data_ptr(), K, W.data_ptr(), K, logits.data_ptr(), G, B, G, K, 1, 0, bias.data_ptr(), None)
    lib.sca_recon_rows(0, logits.data_ptr(), G, None, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), None, sf.data_ptr(), cond.data_ptr(), tab_theta.data_ptr(), tab_lte.data_ptr(), B, G, C, min(16, max(1, B // 512)), row_ll.data_ptr(), dthetaT.data_ptr(), dbias.data_ptr(), None, 1, None, None, None)
def table():
    lib.sca_nb_gamma_table(tab_theta.data_ptr(), C, G, gam.data_ptr(), None)
def fused():
    rc = lib.sca_out_nb_fused(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), None, sf.data_ptr(), cond.data_ptr(), tab_theta.data_ptr(), gam.data_ptr(), B, G, K, C, logits.data_ptr(), G, ovf.data_ptr(), G, row_ll.data_ptr(), dthetaT.data_ptr(), dbias.data_ptr(), 1, None)
    assert rc == 0, rc
table()
for name, fn in (("unfused gemm+recon", unfused), ("gamma table", table), ("fused", fused)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print("%-20s %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))

nwg = (B + 15) // 16
st = torch.zeros(nwg, 12, dtype=torch.int64, device="cuda")
lib.sca_out_nb_fused_debug_stamps(st.data_ptr())
fused(); torch.cuda.synchronize()
lib.sca_out_nb_fused_debug_stamps(None)
s = st.cpu().double() / 100.0       # 100 MHz -> us
t0 = s[:, 0].min()
names = ["scalars+hA", "mfma", "store+stats", "wait+barrier", "rowstats", "nnz", "barrier", "dense1", "dense2", "wait+barrier", "patch+flush"]
d = (s[:, 1:] - s[:, :-1])
print("workgroup start spread %.1f us, end %.1f .. %.1f us" % (float(s[:, 0].max() - t0), float(s[:, 11].min() - t0), float(s[:, 11].max() - t0)))
for i, nm in enumerate(names):
    print("  %-14s mean %.2f  median %.2f  max %.2f us" % (nm, float(d[:, i].mean()), float(d[:, i].median()), float(d[:, i].max())))
