"""Timing of the row-block MMD (sca_mmd_rows_fwd_bwd) against the symmetric whole-minibatch kernel (sca_mmd_fwd_bwd)
for data-parallel shapes: `world` ranks x `n` local cells, Z=10, 20 conditions.  One GPU; prints one line per shape."""
import sys

import torch

sys.path.insert(0, ".")
from scarches_amd import _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    L = _lib.load()
    D, C = 10, 20
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    for world, n in [(1, 1024), (1, 4096), (1, 8192), (1, 16384), (2, 1024), (8, 128), (8, 1024), (2, 8192), (8, 8192)]:
        Bg = world * n
        z = torch.randn(Bg, D, device="cuda", generator=g)
        c = torch.sort(torch.randint(0, C, (Bg,), device="cuda", generator=g).to(torch.int32))[0].contiguous()
        Wt = torch.empty(C * C, device="cuda")
        ns, npart = L.sca_mmd_num_slabs(Bg, D), L.sca_mmd_num_partials(Bg, D)
        part = torch.zeros(npart, dtype=torch.float64, device="cuda")
        slabs = torch.empty(ns * Bg * D, device="cuda")
        full = timeit(lambda: L.sca_mmd_fwd_bwd(z.data_ptr(), D, c.data_ptr(), Bg, D, C, -1, 1.0, Wt.data_ptr(), None,
                                                part.data_ptr(), slabs.data_ptr(), Bg * D, st))
        v_full = part.sum().item()
        nsr, npr = L.sca_mmd_rows_num_slabs(Bg, n, D), L.sca_mmd_rows_num_partials(Bg, n, D)
        partr = torch.zeros(npr, dtype=torch.float64, device="cuda")
        slabr = torch.empty(nsr * n * D, device="cuda")
        rows = timeit(lambda: L.sca_mmd_rows_fwd_bwd(z.data_ptr(), D, c.data_ptr(), Bg, D, C, -1, 1.0, Wt.data_ptr(),
                                                     None, 0, n, partr.data_ptr(), slabr.data_ptr(), n * D, st))
        v = 0.0
        for r in range(world):
            L.sca_mmd_rows_fwd_bwd(z.data_ptr(), D, c.data_ptr(), Bg, D, C, -1, 1.0, Wt.data_ptr(), None, r * n, n,
                                   partr.data_ptr(), None, n * D, st)
            v += partr.sum().item()
        print(f"world {world} x {n:5d} local cells (global {Bg:6d}): whole minibatch on every rank {full:8.1f} us, "
              f"row block {rows:8.1f} us ({n * Bg / rows * 1e-3:.1f} G ordered pairs/s), "
              f"sum of blocks / whole value - 1 = {v / v_full - 1:+.1e}", flush=True)


if __name__ == "__main__":
    main()
