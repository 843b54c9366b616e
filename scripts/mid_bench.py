"""Dev probe: time the fused middle kernels alone (HIP events), dropout on/off, a few batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_trvae_params, epoch_batches
from scarches_amd import _lib
from scarches_amd.engine import CVAEEngine
from scarches_amd.synth import make_synthetic_csr

def phases(L, rf, st, tag):
    import ctypes
    try:
        fn = L.sca_midd_debug
    except AttributeError:
        return
    if True:
        buf = (ctypes.c_longlong * 64)()
        L.sca_mid_fwd(rf, st); torch.cuda.synchronize()
        L.sca_mid_bwd(RB[0], st); torch.cuda.synchronize()
        fn(buf)
        t = [buf[i] for i in range(64)]
        names = ["prefetch", "first slabs", "first epilogue", "enc1 gemm", "enc1 epilogue", "heads gemm", "latent",
                 "dec0 gemm", "dec0 epilogue", "dec1 gemm", "dec1 epilogue"]
        marks = [t[15]] + t[0:11]
        print(tag, "fwd phases (us):", {n: round((marks[i + 1] - marks[i]) / 100.0, 2) for i, n in enumerate(names)},
              "total", round((marks[-1] - marks[0]) / 100.0, 2), flush=True)
        print(tag, "last block: fwd start +%.2f us, end +%.2f us (block 0 ends +%.2f); bwd start +%.2f, end +%.2f (block 0 ends +%.2f)" % (
            (t[16] - t[15]) / 100., (t[17] - t[15]) / 100., (t[10] - t[15]) / 100., (t[48] - t[32]) / 100., (t[49] - t[32]) / 100., (t[43] - t[32]) / 100.), flush=True)
        bn = ["slabs in", "wait", "dec1 epi", "dec1 gemm", "dec0 epi", "dec0 gemm", "latent", "heads gemm", "enc1 epi", "enc1 gemm", "first epi"]
        bm = t[32:44]
        print(tag, "bwd phases (us):", {n: round((bm[i + 1] - bm[i]) / 100.0, 2) for i, n in enumerate(bn)},
              "total", round((bm[-1] - bm[0]) / 100.0, 2), flush=True)
        return
    buf = (ctypes.c_longlong * 64)()
    L.sca_mid_fwd(rf, st); torch.cuda.synchronize()
    fn(buf)
    t = [buf[i] for i in range(16)]
    # marks of mid_fwd_kernel (SCA_MID_DEBUG build, H = [128, 128]): 15 start, 0 first prefetch / input tile, then two per
    # chain position (after the GEMM or slab load, after the epilogue / latent block)
    names = ["prefetch", "first slabs", "first epilogue", "enc1 gemm", "enc1 epilogue", "heads gemm", "latent",
             "dec0 gemm", "dec0 epilogue", "dec1 gemm", "dec1 epilogue"]
    marks = [t[15]] + t[0:11]
    tt = [buf[i] for i in range(64)]
    mhz = (tt[32 + 10] - tt[32 + 15]) / max(1, (tt[10] - tt[15])) * 100.0
    print(tag, "shader clock MHz over the kernel:", round(mhz), "| last gemm, first chunk (us): entry->barrier..store..barrier..load issue..mfma done",
          [round((tt[i + 1] - tt[i]) / 100.0, 2) for i in range(16, 20)], flush=True)
    print(tag, "fwd phases (us):", {n: round((marks[i + 1] - marks[i]) / 100.0, 2) for i, n in enumerate(names)},
          "total", round((marks[-1] - marks[0]) / 100.0, 2), flush=True)

RB = []
dev = torch.device("cuda", 0)
G, C, Z, H = 2000, 20, 10, [int(x) for x in os.environ.get("MID_H", "128,128").split(",")]
indptr, indices, data, cond_all, _ = make_synthetic_csr(50_000, G, C, dev, seed=1000)
P = build_trvae_params(G, C, H, Z)
gen = torch.Generator(device=dev); gen.manual_seed(1)
for B in [int(x) for x in os.environ.get('MID_BS', '128,4096,16384').split(',')]:
    for p in (0.0, 0.05):
        spec = dict(G=G, C=C, H=H, Z=Z, recon="nb", use_bn=False, use_ln=True, use_mmd=True, mmd_on="z",
                    boundary=None, beta=1.0, dr_rate=p, seed=2020)
        eng = CVAEEngine(spec, P, dev)
        eng.bind_csr(indptr, indices, data)
        ids, cs = epoch_batches(cond_all, 2, B, gen)
        eng.begin_step(0.5, 1e-3)
        import os
        masks = None
        if os.environ.get("MID_MASKS") == "1" and p > 0:
            masks = {ly.mask_key: (torch.rand(B, ly.n_out, device=dev) > p).to(torch.uint8) for ly in eng.enc + eng.dec}
        eng.forward_backward(ids[0], cs[0], B, masks=masks)
        torch.cuda.synchronize()
        st = torch.cuda.current_stream().cuda_stream
        L = _lib.load()
        (af, rf), = [v for k, v in eng._mid_cache.items() if k[7] is None]
        (ab, rb), = [v for k, v in eng._mid_cache.items() if k[7] is not None]
        RB[:] = [rb]
        for name, fn, ref in (("fwd", L.sca_mid_fwd, rf), ("bwd", L.sca_mid_bwd, rb)):
            for _ in range(3):
                fn(ref, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(ref, st)
            e1.record()
            torch.cuda.synchronize()
            print(f"B={B:6d} p={p:.2f} mid_{name}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
        phases(L, rf, st, f"B={B} p={p}")

