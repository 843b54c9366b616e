"""Host-side driver of the HIP conditional-VAE step (trVAE family).

Owns the flat parameter / gradient / Adam-state buffers in HBM, the per-batch workspaces, and the launch order of the
kernels behind the C ABI (include/scarches_hip.h).  PyTorch is used for device memory, streams and (elsewhere)
torch.distributed only; all arithmetic of the step runs in libscarches_hip.so.  There is no CPU fallback.

Reference path reproduced (see the kernel files for line-level citations):
  scarches/models/trvae/trvae.py:116-157 (forward), trainers/trvae/unsupervised.py:62-64 (loss composition),
  trainers/trvae/trainer.py:221-223,262-280 (Adam + iteration), models/base/_base.py:409-455 (get_latent / get_y).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import os

import torch

from . import _lib

ALIGN = 64  # floats (256 B) - every tensor starts on its own 256-byte boundary inside the flat buffers


def _round_up(n, a):
    return (n + a - 1) // a * a


class _Layer:
    __slots__ = ("tag", "w", "b", "wc", "n_in", "n_out", "rng_stream", "mask_key", "bn")

    def __init__(self, tag, w, b, wc, n_in, n_out, rng_stream, mask_key, bn=None):
        self.tag, self.w, self.b, self.wc = tag, w, b, wc
        self.n_in, self.n_out, self.rng_stream, self.mask_key = n_in, n_out, rng_stream, mask_key
        self.bn = bn        # state-dict prefix of the BatchNorm1d after this Linear (use_bn), e.g. "encoder.FC.N0"


_QUERIES = {"sca_gemm_kk_supported", "sca_gemm_kk_splitk_heuristic", "sca_mid_max_width", "sca_grad_tail_max_jobs",
            "sca_gemm_group_max_jobs", "sca_gemm_splitk_heuristic", "sca_mmd_num_slabs", "sca_mmd_num_partials",
            "sca_mmd_rows_num_slabs", "sca_mmd_rows_num_partials", "sca_hsic_workspace_floats",
            "sca_proto_workspace_bytes", "sca_abi_version", "sca_device_name", "sca_masked_dw_slabs",
            "sca_masked_dx_slabs", "sca_masked_max_cols", "sca_masked_max_width", "sca_mmd_prepared_supported", "sca_mmd_select_form",
            "sca_masked_fwd_tiles_per_wg"}


# whole-minibatch MMD in two row blocks around the likelihood kernel (fused trVAE step): from this many cells on, first
# block = this fraction of the rows (0 or 1: one launch).  Off by default: measured on MI355X at 8192 cells the second
# block beside the output layer's dX / dW GEMMs costs those one-workgroup-per-CU kernels more than the likelihood kernel
# gains (0.684 ms per step in one launch; 0.794 / 0.779 / 0.763 ms with 0.4 / 0.5 / 0.6).
MMD_SPLIT_MIN = int(os.environ.get("SCA_MMD_SPLIT_MIN", 4096))
MMD_SPLIT_FRAC = float(os.environ.get("SCA_MMD_SPLIT_FRAC", 0.0))
MMD_AFTER_OUT = os.environ.get("SCA_MMD_AFTER_OUT", "auto")      # "0" / "1" / "auto": see _fused_path
MMD_AFTER_OUT_MAX = int(os.environ.get("SCA_MMD_AFTER_OUT_MAX", 6144))
MMD_AFTER_OUT_MAX_GRAPH = int(os.environ.get("SCA_MMD_AFTER_OUT_MAX_GRAPH", 8192))   # while a HIP graph is captured
GRAPH_PREFETCH = os.environ.get("SCA_GRAPH_PREFETCH", "0") == "1"     # look-ahead gather inside captured graphs too
GRAPH_MAX_B = int(os.environ.get("SCA_GRAPH_MAX_B", 8192))      # train_step(use_graph="auto"): HIP graph up to here
MMD_PAD, MMD_PAD_AFTER_OUT = 40000, 24000       # sca_mmd_reserve_lds per fork point (SCA_MMD_LDS_PAD overrides both)


class _RecordingLib:
    """The C ABI as seen by the engine: every launch goes straight to libscarches_hip.so and, while the engine records
    a launch plan (CVAEEngine.train_step), is also appended to it as (function, argument tuple)."""

    def __init__(self, lib, eng):
        self._lib, self._eng = lib, eng

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in _QUERIES:
            raw, eng = fn, self._eng

            def fn(*args):
                if eng._rec is not None:
                    eng._rec.append((raw, args, True))
                return raw(*args)
        setattr(self, name, fn)
        return fn


def _recon_rows_per_block(B):
    """Rows a likelihood workgroup walks (theta gradients stay in registers across them): ~512 workgroups = two per
    CU, at most 16 rows.  Measured inside the cfg2 step (scripts/ab_env.sh, ms/step): B=1024 0.305 / 0.287 / 0.293 with
    1 / 2 / 4 rows, B=2048 0.326 / 0.313 / 0.335 with 2 / 4 / 8, B=4096 0.431 / 0.409 / 0.396 / 0.444 with 2 / 4 / 8 / 16,
    B=8192 0.695 / 0.693 / 0.673 / 0.674 with 2 / 4 / 8 / 16.  SCA_RECON_RPB overrides it for A/B runs.
    Round 3, the step replayed as a HIP graph (MMD forked behind the output GEMM, the likelihood kernel placed first):
    8192 cells 0.603 / 0.563 / 0.565 / 0.552 / 0.552 / 0.559 / 0.601 ms with 8 / 16 / 18 / 20 / 22 / 24 / 32 rows (three
    runs each for 16 and 20) - 20 rows there; 7168 cells 0.493 / 0.492 / 0.495 with 14 / 16 / 18, 6144: 0.422 / 0.428 with
    12 / 15, 4096: 8 - 9 rows, so the rule below stays elsewhere; launch plans at 8192 cells want 16 (0.574 vs 0.601
    with 20): the MMD is forked ahead of the likelihood kernel there."""
    v = os.environ.get("SCA_RECON_RPB")
    if v:
        return max(1, int(v))
    if B > 7680 and B <= GRAPH_MAX_B and torch.cuda.is_current_stream_capturing():
        return 20
    return min(16, max(1, B // 512))


class CVAEEngine:
    """Forward / backward / Adam of a trVAE-style conditional VAE on one MI355X."""

    # weight-gradient GEMMs with at least this many output elements take the register-direct k-major kernel (one
    # workgroup per CU); smaller ones stay in the grouped launch of the LDS-staged kernel
    KK_MIN_MN = int(os.environ.get("SCA_GEMM_KK_MIN_MN", 65536))

    def __init__(self, spec: dict, named_params: "OrderedDict[str, torch.Tensor]", device):
        self._rec = None
        self._ws_gen = 0
        self._plans = {}
        self._adam_after_tail = None    # (weight_decay, wd_overrides): train_step on one device queues Adam on the aux stream
        self._dp_dry = False            # True while a warm-up / recording pass runs: data-parallel gathers stay local
        self._adam_done = False
        self._dp_split = False          # True while a data-parallel step is issued: all-reduce between tail and Adam
        self._dp_world, self._dp_grad_weight = 1, None
        # split-K cap of the two GEMMs whose slabs the fused middle kernels sum tile by tile: the planner's 32 slabs at
        # <= 1024 cells cost the 16-cell tiles more (four rounds of slab loads) than the GEMMs gain - cfg2 step, graph
        # launched: 128 cells 0.1446 - 0.1485 ms with 32, 0.1413 with 16, 0.1445 with 8, 0.161 with 4; 1024 cells 0.1954 /
        # 0.188 / 0.190 / 0.207 (SCA_MID_MAX_SPLIT overrides)
        self._mid_max_split = int(os.environ.get("SCA_MID_MAX_SPLIT") or 16)
        self.lib = _RecordingLib(_lib.load(), self)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HipLibraryError("scarches_amd runs on a HIP device only (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        s = dict(spec)
        self.G, self.C, self.H, self.Z = int(s["G"]), int(s["C"]), list(s["H"]), int(s["Z"])
        self.recon = s["recon"]
        self.use_bn, self.use_ln = bool(s.get("use_bn", False)), bool(s.get("use_ln", True))
        # BatchNorm takes precedence over LayerNorm, as in the reference (modules.py:78-81)
        self.norm = 1 if (self.use_ln and not self.use_bn) else 0
        self.freeze = bool(s.get("freeze", False))
        self.dr_rate = float(s.get("dr_rate", 0.0))
        self.use_mmd = bool(s.get("use_mmd", False))
        self.mmd_on = s.get("mmd_on", "z")
        self.boundary = s.get("boundary", None)
        self.beta = float(s.get("beta", 1.0))
        self.seed = int(s.get("seed", 2020))
        assert self.recon in ("nb", "zinb", "mse", "poisson")
        # the dispersion parameter theta only exists for the negative-binomial family (trvae.py:89-92, scpoli.py:83-88)
        self.has_theta = self.recon in ("nb", "zinb")
        assert len(self.H) >= 1, "at least one hidden layer is required"
        # architecture family: "trvae" (MLP decoder, one-hot conditions), "expimap" (single masked linear decoder
        # layer + extension terms, scarches/models/expimap/expimap.py:44-171)
        self.arch = s.get("arch", "trvae")
        assert self.arch in ("trvae", "expimap", "scpoli")
        self.n_ext, self.n_ext_m = int(s.get("n_ext", 0)), int(s.get("n_ext_m", 0))
        self.Zx = self.n_ext + self.n_ext_m          # expand_{mean,var}_encoder width (ExtEncoder, modules.py:198-215)
        self.Zt = self.Z + self.Zx                   # width of the sampled latent
        self.use_hsic = bool(s.get("use_hsic", False)) and self.n_ext > 0
        self.hsic_ova = bool(s.get("hsic_one_vs_all", False))
        self.hsic_beta = float(s.get("hsic_beta", 1.0))
        if self.arch == "expimap":
            assert self.recon in ("nb", "mse")
            # MaskedLinearDecoder last layer (models/expimap/modules.py:96-127): softmax (nb default), identity (mse
            # default), softplus, exp, relu; the non-softmax heads give an elementwise mean without the size factor
            last = s.get("decoder_last_layer") or ("softmax" if self.recon == "nb" else "identity")
            if last not in ("softmax", "identity", "relu", "softplus", "exp"):
                raise ValueError("Unrecognized last layer.")
            if self.recon == "mse" and last == "softmax":
                raise ValueError("Can't specify softmax last layer with mse loss.")
            self.last_layer = last
            self.last_act = {"identity": 0, "relu": 1, "softplus": 2, "exp": 3}.get(last, -1)
            # library-size encoder (expimap.py:164-171): only consulted with nb + softmax (expimap.py:186-187)
            self.use_l_encoder = bool(s.get("use_l_encoder", False))
            self.lenc_active = self.use_l_encoder and self.recon == "nb" and last == "softmax"
            self.use_mmd = False
        # scPoli: conditions enter as dense embedding vectors (hstack of nn.Embedding lookups, one table per
        # covariate); theta is indexed by the combined condition (scarches/models/scpoli/scpoli.py:99-101,137,161-176)
        self.cond_embed = self.arch == "scpoli"
        self.emb = []
        if self.cond_embed:
            self.E_list, self.n_conds = [int(e) for e in s["E"]], [int(n) for n in s["n_conds"]]
            self.E = sum(self.E_list)
            self.inject = set(s.get("inject", ["encoder", "decoder"]))
            mn = s.get("max_norm", 1.0)
            self.max_norm = 0.0 if mn is None else float(mn)
            self.use_mmd = False
            self.emb = [(f"embeddings.{i}.weight", n, e) for i, (n, e) in enumerate(zip(self.n_conds, self.E_list))]

        # ---- flat parameter storage ------------------------------------------------------------------------
        # mean/log_var (+ expand) heads are stored back to back so that all run as ONE [2*Zt, H] GEMM whose output
        # row is [mean | expand_mean | log_var | expand_var] = [means(Zt) | log_vars(Zt)]
        order = list(named_params.keys())
        xw = self.Zx > 0
        fused_w = (["encoder.mean_encoder.weight"] + (["encoder.expand_mean_encoder.weight"] if xw else [])
                   + ["encoder.log_var_encoder.weight"] + (["encoder.expand_var_encoder.weight"] if xw else []))
        fused_b = [k[:-6] + "bias" for k in fused_w]
        fused = fused_w + fused_b
        # expiMap's library-size encoder: its two 1 x 128 heads are stored back to back as well ([2, 128] GEMM)
        lenc = getattr(self, "lenc_active", False) or getattr(self, "use_l_encoder", False)
        fused_lw = ["l_encoder.mean_encoder.weight", "l_encoder.log_var_encoder.weight"] if lenc else []
        fused_lb = [k[:-6] + "bias" for k in fused_lw]
        order = [k for k in order if k not in fused and k not in fused_lw and k not in fused_lb]
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, tuple] = {}
        off = 0
        for k in order:
            self.offsets[k] = off
            self.shapes[k] = tuple(named_params[k].shape)
            off = _round_up(off + named_params[k].numel(), ALIGN)
        for pair in (fused_w, fused_b, fused_lw, fused_lb):
            for k in pair:
                self.offsets[k] = off
                self.shapes[k] = tuple(named_params[k].shape)
                off += named_params[k].numel()
            off = _round_up(off, ALIGN)
        self.n_flat = off
        self.names = list(named_params.keys())
        dev = self.device
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for k in self.names:
            o, n = self.offsets[k], int(np.prod(self.shapes[k]))
            self.p[k] = self.flat_p[o:o + n].view(self.shapes[k])
            self.g[k] = self.flat_g[o:o + n].view(self.shapes[k])
            self.p[k].copy_(named_params[k].detach().to(torch.float32))
        self.trainable = {k: True for k in self.names}
        self.adam_t = 0

        # ---- layer table -----------------------------------------------------------------------------------
        H, G, Z = self.H, self.G, self.Z
        has_c = self.C > 0
        has_c_enc = has_c_dec = has_c
        if self.cond_embed:
            has_c_enc = self.E > 0 and "encoder" in self.inject
            has_c_dec = self.E > 0 and "decoder" in self.inject
        self.enc: List[_Layer] = []
        for i, h in enumerate(H):
            if i == 0:
                self.enc.append(_Layer(f"enc{i}", "encoder.FC.L0.expr_L.weight", "encoder.FC.L0.expr_L.bias",
                                       "encoder.FC.L0.cond_L.weight" if has_c_enc else None, G, h, 10 + i,
                                       f"enc{i}"))
            else:
                self.enc.append(_Layer(f"enc{i}", f"encoder.FC.L{i}.weight", f"encoder.FC.L{i}.bias", None,
                                       H[i - 1], h, 10 + i, f"enc{i}"))
        Hd = list(reversed(H))
        self.dec: List[_Layer] = []
        if self.arch == "expimap":
            # one linear layer z[Zt] -> G; its weight is assembled per step from the (masked) pieces
            self.xp_parts = [("decoder.L0.expr_L.weight", Z, s.get("mask"))]
            if self.n_ext_m:
                self.xp_parts.append(("decoder.L0.ext_L_m.weight", self.n_ext_m, s.get("ext_mask")))
            if self.n_ext:
                self.xp_parts.append(("decoder.L0.ext_L.weight", self.n_ext, None))
            self.xp_masks = [None if m is None else m.detach().to(dev, torch.float32).contiguous()
                             for (_, _, m) in self.xp_parts]
            for (k, w, m), md in zip(self.xp_parts, self.xp_masks):
                assert self.shapes[k] == (G, w), (k, self.shapes[k])
                assert md is None or tuple(md.shape) == (G, w), "mask must be [G, n_terms] (MaskedLinear.mask layout)"
            self.xp_wc = "decoder.L0.cond_L.weight" if has_c else None
            self.out_w = self.out_b = None
            self.masked = self._masked_meta()
        else:
            self.dec.append(_Layer("dec0", "decoder.FirstL.L0.expr_L.weight", None,
                                   "decoder.FirstL.L0.cond_L.weight" if has_c_dec else None, Z, Hd[0], 40, "dec0"))
            for i in range(1, len(Hd)):
                self.dec.append(_Layer(f"dec{i}", f"decoder.HiddenL.L{i}.weight", None, None, Hd[i - 1], Hd[i],
                                       40 + i, f"dec{i}"))
            if self.recon in ("mse", "poisson"):      # scPoli's poisson head is also called recon_decoder (scpoli.py:607)
                self.out_w, self.out_b = "decoder.recon_decoder.0.weight", "decoder.recon_decoder.0.bias"
            else:
                self.out_w, self.out_b = "decoder.mean_decoder.0.weight", "decoder.mean_decoder.0.bias"
        # BatchNorm1d(affine) after every hidden Linear when use_bn: parameters <prefix>.weight/.bias live in the flat
        # buffer, the running statistics are buffers handed in through spec["buffers"] (updated in place)
        self.buffers: Dict[str, torch.Tensor] = {}
        if self.use_bn:
            for i, ly in enumerate(self.enc):
                ly.bn = f"encoder.FC.N{i}"
            for i, ly in enumerate(self.dec):
                ly.bn = "decoder.FirstL.N0" if i == 0 else f"decoder.HiddenL.N{i}"
            given = s.get("buffers") or {}
            for ly in self.enc + self.dec:
                for suffix, init in (("running_mean", 0.0), ("running_var", 1.0)):
                    k = f"{ly.bn}.{suffix}"
                    t = given.get(k)
                    t = torch.full((ly.n_out,), init) if t is None else t.detach()
                    self.buffers[k] = t.to(dev, torch.float32).contiguous().clone()
                k = f"{ly.bn}.num_batches_tracked"
                t = given.get(k)
                self.buffers[k] = (torch.zeros((), dtype=torch.long) if t is None else t.detach()).to(dev).clone()
        # operand shapes the kernels and their grids assume, checked on the host before anything is launched
        def _expect(k, shape):
            if k is not None and self.shapes.get(k) != tuple(shape):
                raise ValueError("parameter %r has shape %r, the architecture (G=%d, H=%r, Z=%d, C=%d) needs %r"
                                 % (k, self.shapes.get(k), G, H, self.Zt, self.C, tuple(shape)))
        for ly in self.enc + self.dec:
            _expect(ly.w, (ly.n_out, ly.n_in))
            _expect(ly.b, (ly.n_out,))
            _expect(ly.wc, (ly.n_out, self.E if self.cond_embed else self.C))
            if ly.bn:
                _expect(ly.bn + ".weight", (ly.n_out,))
                _expect(ly.bn + ".bias", (ly.n_out,))
        if self.out_w:
            _expect(self.out_w, (G, Hd[-1]))
            _expect(self.out_b, (G,))
            if self.recon == "zinb":
                _expect("decoder.dropout_decoder.weight", (G, Hd[-1]))
                _expect("decoder.dropout_decoder.bias", (G,))
        if self.has_theta and self.C > 0:
            _expect("theta", (G, self.C))
        for k in fused_w:
            _expect(k, (self.Zx if "expand" in k else Z, H[-1]))
        for k, n, e in self.emb:
            _expect(k, (n, e))
        self.lenc = None
        if lenc:
            if self.use_bn:
                raise NotImplementedError("use_l_encoder=True together with use_bn=True is not built in the HIP path")
            Hl_ = self.shapes["l_encoder.FC.L0.expr_L.weight"][0]       # 128 in the reference (expimap.py:165)
            self.lenc = _Layer("lenc0", "l_encoder.FC.L0.expr_L.weight", "l_encoder.FC.L0.expr_L.bias",
                               "l_encoder.FC.L0.cond_L.weight" if has_c else None, G, Hl_, 60, "lenc0")
            _expect(self.lenc.w, (Hl_, G))
            _expect(self.lenc.b, (Hl_,))
            _expect(self.lenc.wc, (Hl_, self.C))
            for k in fused_lw:
                _expect(k, (1, Hl_))
            lw, lb = self.offsets[fused_lw[0]], self.offsets[fused_lb[0]]
            self.lheads_w = self.flat_p[lw:lw + 2 * Hl_].view(2, Hl_)
            self.lheads_b = self.flat_p[lb:lb + 2]
            self.lheads_gw = self.flat_g[lw:lw + 2 * Hl_].view(2, Hl_)
            self.lheads_gb = self.flat_g[lb:lb + 2]
            self.lheads_names = fused_lw + fused_lb
            self.zero_hyper = torch.zeros(8, dtype=torch.float32, device=dev)     # KL weight 0: the library latent
                                                                                  # carries no KL term (expimap.py:198)
        Zt = self.Zt
        ow, ob = self.offsets[fused_w[0]], self.offsets[fused_b[0]]
        self.heads_w = self.flat_p[ow:ow + 2 * Zt * H[-1]].view(2 * Zt, H[-1])
        self.heads_b = self.flat_p[ob:ob + 2 * Zt]
        self.heads_gw = self.flat_g[ow:ow + 2 * Zt * H[-1]].view(2 * Zt, H[-1])
        self.heads_gb = self.flat_g[ob:ob + 2 * Zt]
        self.heads_names = fused

        # ---- small persistent device state -----------------------------------------------------------------
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        # pinned staging ring: an async H2D copy reads host memory when it executes, so each in-flight
        # update needs its own slot (an event per slot guards reuse)
        self._hyper_host = torch.zeros(8, dtype=torch.float32)
        self._hyper_host[6] = 1.0       # epoch-log weight of a minibatch (set_hyper(log_weight=...) under DP)
        self._hyper_ring = torch.zeros(64, 8, dtype=torch.float32).pin_memory()
        self._hyper_ev = [None] * 64
        self._hyper_i = 0
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=dev)      # mixed into the Philox streams
        self.losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self.accum = torch.zeros(5, dtype=torch.float64, device=dev)
        self.csr = None
        self.Bcap = 0
        # side stream: MMD runs beside the decoder, every dW / bias / condition-weight gradient beside the dX chain
        self.aux = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("SCA_AUX_PRIO", "0")))
        # output-layer weight gradient beside the dX chain
        self.aux2 = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("SCA_AUX2_PRIO", "0")))
        self._lanes_used = set()
        self._side_deps = set()
        self._events = [torch.cuda.Event() for _ in range(128)]
        self._ev_i = 0
        self._aux_q = []
        self._rec = None                # launch-plan recorder (train_step)
        self._pf = None                 # minibatch that ws["x"] already holds (train_step(next_rows=...))
        self._pf_flags = (False, False) # (skip the gather at the head of the step, gather the next one at its end)
        self._rows_next = None
        self.overlap = True
        self._dp = None                 # set_dp(): rank / world / exact-MMD switch of data-parallel runs
        self._dp_ws = {}
        self._dp_n_global = None
        self.set_hyper(alpha_kl=1.0, lr=1e-3, eps=0.01, clip=0.0)
        # fused middle (fused_mid.hip): trVAE family with LayerNorm / no norm, NB or MSE, narrow hidden layers
        wmax = self.lib.sca_mid_max_width()
        self.use_fused_mid = (self.arch == "trvae" and not self.use_bn and self.recon in ("nb", "mse")
                              and len(self.enc) - 1 <= 4 and 1 <= len(self.dec) <= 4
                              and max(self.H) <= wmax and 3 * self.Zt <= wmax and bool(s.get("fused_mid", True)))

    # ------------------------------------------------------------------------------------------ configuration
    def set_hyper(self, alpha_kl=None, lr=None, eps=None, clip=None, t=None, log_weight=None):
        h = self._hyper_host
        if log_weight is not None:
            h[6] = float(log_weight)
        if alpha_kl is not None:
            h[0] = float(alpha_kl)
        if lr is not None:
            h[1] = float(lr)
        if t is not None:
            h[2] = 1.0 - 0.9 ** t
            h[3] = math.sqrt(1.0 - 0.999 ** t)
        if eps is not None:
            h[4] = float(eps)
        if clip is not None:
            h[5] = float(clip)
        i = self._hyper_i = (self._hyper_i + 1) % 64
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()
        self._hyper_ring[i].copy_(h)
        self.hyper.copy_(self._hyper_ring[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._hyper_ev[i] = ev

    # `noise_off = True` replaces the reparameterisation noise by zeros wherever no eps is injected (z = mean): the
    # reference samples also in eval mode (_base.py:391-407), so fixtures of its sampled outputs (classify,
    # get_recon_loss, add_new_cell_type) are taken with the noise switched off on both sides
    noise_off = False

    def _eps_or_zero(self, eps, B):
        if eps is not None or not self.noise_off:
            return eps
        z = getattr(self, "_zero_eps", None)
        if z is None or z.shape[0] < B:
            z = self._zero_eps = torch.zeros(max(B, 1024), self.Zt, dtype=torch.float32, device=self.device)
        return z

    def bind_csr(self, indptr: torch.Tensor, indices: torch.Tensor, data: torch.Tensor):
        """CSR matrix resident in HBM: indptr int64 [N+1], indices int32 [nnz], data float32 [nnz]."""
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32 and data.dtype == torch.float32
        assert indptr.device == self.device and indices.device == self.device and data.device == self.device
        self.csr = (indptr.contiguous(), indices.contiguous(), data.contiguous())

    def _masked_meta(self):
        """Tile-skipping plan of the masked decoder (csrc/masked_gemm.hip): genes in tiles of 32, per tile the sorted
        gene-program columns that are not fully masked for its genes, padded to an even count (row length Kmax: a
        multiple of 8) with pointers at the zero column Zt.  Built once from the binary pattern of the masks (MaskedLinear.mask never changes after
        construction, modules.py:9-25); None = use the dense GEMMs (soft-mask / unmasked models, a plan that would
        skip too little, or more columns than the kernels hold)."""
        if os.environ.get("SCA_MASKED_GEMM", "1") == "0" or self.xp_masks[0] is None:
            return None
        G, Zt = self.G, self.Zt
        pat = np.concatenate([np.ones((G, w), bool) if md is None else (md.cpu().numpy() != 0)
                              for (_, w, _), md in zip(self.xp_parts, self.xp_masks)], axis=1)
        NT = (G + 31) // 32
        pad = np.zeros((NT * 32, Zt), bool)
        pad[:G] = pat
        active = pad.reshape(NT, 32, Zt).any(1)                    # [NT][Zt]
        k = active.sum(1)
        k_even = (k + 1) & ~1
        Kmax = max((int(k_even.max()) + 7) & ~7, 8)
        skipped = 1.0 - float(k_even.sum()) / (NT * Zt)
        min_skip = float(os.environ.get("SCA_MASKED_MIN_SKIP", 0.3))
        if Kmax > self.lib.sca_masked_max_cols() or Zt > self.lib.sca_masked_max_width() or skipped < min_skip:
            return None
        cols = np.full((NT, Kmax), Zt, np.int32)
        for t in range(NT):
            c = np.nonzero(active[t])[0]
            cols[t, :len(c)] = c
        dev = self.device
        return dict(NT=NT, Kmax=Kmax, skipped=skipped, tile_k=torch.tensor(k_even.astype(np.int32), device=dev),
                    tile_cols=torch.tensor(cols, device=dev), n=NT * Kmax * 32)

    def set_trainable(self, names):
        names = set(names)
        for k in self.names:
            self.trainable[k] = k in names
        if self.lenc is not None and not self.lenc_active:
            # an unused library-size encoder (expimap.py:186: only nb + softmax consults it) never receives a
            # gradient in the reference, so torch's Adam skips it - weight decay included
            for k in self.names:
                if k.startswith("l_encoder."):
                    self.trainable[k] = False

    def _out_fused_ok(self):
        """Output layer + NB likelihood as ONE launch (csrc/out_nb_fused.hip): the fused trVAE step with the plain NB
        head on a CSR matrix, shapes the kernel holds in registers (G <= 2048, last hidden width a multiple of 16)."""
        # OPT-IN (SCA_OUT_FUSED=1): parity-green, but measured slower than the GEMM + likelihood pair on MI355X
        # (246 vs 177 us alone at 8192 cells)
        return bool(os.environ.get("SCA_OUT_FUSED", "0") == "1"
                    and self.use_fused_mid and self.recon == "nb" and self.arch == "trvae" and self.out_w is not None
                    and self.lib.sca_out_nb_fused_supported(self.G, self.dec[-1].n_out))

    # ------------------------------------------------------------------------------------------ workspaces
    def _invalidate(self):
        """Drop everything that holds raw workspace addresses (recorded launch plans, HIP graphs, cached kernel
        argument blocks): called whenever a workspace buffer is (re)allocated."""
        if getattr(self, "_plans", None):
            # a HIP graph must not be destroyed (nor its buffers freed) while its last replay is still running
            torch.cuda.synchronize(self.device)
        self._plans = {}
        self._mid_cache = {}
        self._gg_cache = {}
        self._tail_cache = {}
        self._ws_gen = getattr(self, "_ws_gen", 0) + 1

    def _ensure(self, B):
        if B <= self.Bcap:
            return
        self._invalidate()
        dev, f32 = self.device, torch.float32
        G, C, Z, H = self.G, max(self.C, 1), self.Zt, self.H
        L = self.lib
        Hmax = max(H)
        xp = self.arch == "expimap"
        ws = {}
        ws["x"] = torch.empty(B, G, dtype=f32, device=dev)
        ws["sf"] = torch.empty(B, dtype=f32, device=dev)
        for ly in self.enc + self.dec:
            ws[ly.tag + ".yn"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
            ws[ly.tag + ".rs"] = torch.empty(B, dtype=f32, device=dev)
            ws[ly.tag + ".h"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
        if self.use_bn:
            ws["bn_acc"] = torch.zeros(2 * Hmax, dtype=torch.float64, device=dev)
            for ly in self.enc + self.dec:
                ws[ly.tag + ".bnstats"] = torch.empty(2, ly.n_out, dtype=f32, device=dev)
        ws["mu_lv"] = torch.empty(B, 2 * Z, dtype=f32, device=dev)
        ws["eps"] = torch.empty(B, Z, dtype=f32, device=dev)
        ws["z"] = torch.empty(B, Z, dtype=f32, device=dev)
        ws["kl_row"] = torch.empty(B, dtype=f32, device=dev)
        ws["row_ll"] = torch.empty(B, dtype=f32, device=dev)
        ws["logits"] = torch.empty(B, G, dtype=f32, device=dev)
        if self.recon == "zinb":
            ws["pi"] = torch.empty(B, G, dtype=f32, device=dev)
        if self.has_theta and C * G <= (1 << 22) and (os.environ.get("SCA_GAMMA_TABLE", "0") == "1"
                                                      or self._out_fused_ok()):
            # (lgamma, digamma) differences of integer counts <= 16 per (condition, gene), rebuilt with the theta tables.
            # OPT-IN for the row kernel (SCA_GAMMA_TABLE=1): measured inside the cfg2 step the scattered reads of the
            # 5.4 MB table cost more than the per-count product loop they replace (0.602 -> 0.616 ms at 8192 cells,
            # 0.340 -> 0.347 at 4096); the fused output kernel needs the table.
            ws["gam_tab"] = torch.empty(C, G, 17, 2, dtype=f32, device=dev)
        if self._out_fused_ok():
            ws["ext_ovf"] = torch.empty(B, G, dtype=f32, device=dev)      # touched only by cells with > 512 counts
        ws["tab_theta"] = torch.empty(C, G, dtype=f32, device=dev)
        ws["tab_lte"] = torch.empty(C, G, dtype=f32, device=dev)
        ws["dthetaT"] = torch.empty(C, G, dtype=f32, device=dev)
        ws["segT"] = torch.empty(C * max(Hmax, 2 * Z, G if xp else 0), dtype=f32, device=dev)
        if self.cond_embed:
            E = max(self.E, 1)
            ws["cvec"] = torch.zeros(B, E, dtype=f32, device=dev)
            ws["dcvec_enc"] = torch.empty(B, E, dtype=f32, device=dev)
            ws["dcvec_dec"] = torch.empty(B, E, dtype=f32, device=dev)
            ws["emb_flags"] = torch.zeros(max(self.n_conds + [1]), dtype=torch.uint8, device=dev)
            ws["proto_dz"] = torch.empty(B, Z, dtype=f32, device=dev)
            ws["proto_raw"] = torch.zeros(2, dtype=f32, device=dev)
        if xp:
            ws["Weff"] = torch.empty(G, Z, dtype=f32, device=dev)      # packed [expr*mask | ext_m*mask | ext]
            ws["dWeff"] = torch.empty(G, Z, dtype=f32, device=dev)
            ws["coffT"] = torch.empty(C, G, dtype=f32, device=dev)     # decoder cond_L.weight transposed
            if self.masked is not None:
                mk = self.masked
                bs = list(range(16, B, 16)) + [B]
                ws["mk_Wc"] = torch.empty(2 * mk["n"], dtype=f32, device=dev)      # dX layout + forward layout
                ws["mk_dWc"] = torch.empty(max(L.sca_masked_dw_slabs(b, mk["NT"]) for b in bs) * mk["n"], dtype=f32,
                                           device=dev)
                ws["mk_dz"] = torch.empty(max(L.sca_masked_dx_slabs(b, mk["NT"]) * b for b in bs) * Z, dtype=f32,
                                          device=dev)
            if self.use_hsic:
                ws["hsic_work"] = torch.empty(L.sca_hsic_workspace_floats(B), dtype=f32, device=dev)
                ws["hsic_dz"] = torch.empty(B, Z, dtype=f32, device=dev)
                ws["hsic_raw"] = torch.zeros(1, dtype=f32, device=dev)
                sides = []
                if not self.hsic_ova:       # hsic(z[:, :-n_ext], z[:, -n_ext:]), expimap.py:205-208
                    sides.append([0] * (Z - self.n_ext) + [1] * self.n_ext)
                else:                       # each unconstrained term vs all the others, expimap.py:209-218
                    shift = self.Z + self.n_ext_m
                    for i in range(self.n_ext):
                        sides.append([0 if k == shift + i else 1 for k in range(Z)])
                ws["hsic_side"] = torch.tensor(sides, dtype=torch.uint8, device=dev)

                def bw(d):          # RBF bandwidth of models/expimap/losses.py:15-21
                    return 1.0 / (2.0 * (2.0 * math.gamma(0.5 * (d + 1)) / math.gamma(0.5 * d)) ** 2)
                self._hsic_sigmas = [(bw(sd.count(0)), bw(sd.count(1))) for sd in sides]
        if self.lenc is not None:
            ly = self.lenc
            ws["lenc0.yn"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
            ws["lenc0.rs"] = torch.empty(B, dtype=f32, device=dev)
            ws["lenc0.h"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
            ws["lenc0.da"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
            for k, w in (("l_mu_lv", 2), ("l_eps", 1), ("l_z", 1), ("l_kl", 1), ("sf_l", 1), ("dsf", 1), ("l_dz", 1),
                         ("l_dheads", 2)):
                ws[k] = torch.empty(B, w, dtype=f32, device=dev)
        ws["da"] = torch.empty(B, max(Hmax, 2 * Z), dtype=f32, device=dev)
        ws["dheads"] = torch.empty(B, 2 * Z, dtype=f32, device=dev)
        ws["dh0"] = torch.empty(B, H[0], dtype=f32, device=dev)
        # split-K slab workspace: the largest S*M*N over every GEMM of the step
        need = 0
        Bsym = -1        # placeholder for "the batch size" in the shape list below
        shapes = []
        for ly in self.enc + self.dec:
            shapes += [(Bsym, ly.n_out, ly.n_in), (ly.n_out, ly.n_in, Bsym), (Bsym, ly.n_in, ly.n_out)]
        shapes += [(Bsym, 2 * Z, H[-1]), (2 * Z, H[-1], Bsym), (Bsym, H[-1], 2 * Z)]
        if self.cond_embed:
            for ly in self.enc + self.dec:
                if ly.wc:
                    shapes += [(Bsym, self.E, ly.n_out), (ly.n_out, self.E, Bsym)]
        if self.lenc is not None:
            Hl_ = self.lenc.n_out
            shapes += [(Bsym, Hl_, G), (Hl_, G, Bsym), (Bsym, 2, Hl_), (2, Hl_, Bsym), (Bsym, Hl_, 2)]
        if xp:
            shapes += [(G, Z, Bsym), (Bsym, Z, G)]
        else:
            Hl = self.dec[-1].n_out
            shapes += [(G, Hl, Bsym), (Bsym, Hl, G)]
        # The split count is not monotone in the batch size (fewer 64-row tiles -> the planner doubles S again), so a
        # ragged last minibatch / validation batch B' < B can need MORE slab floats than B itself: size for the worst
        # case over every B' <= B (the plan only changes at multiples of 16 rows: k-tiles of 16, row tiles of 64).
        cache = {}
        for b in list(range(16, B, 16)) + [B]:
            for (M, N, K) in shapes:
                key = (b if M == Bsym else M, N, b if K == Bsym else K)
                v = cache.get(key)
                if v is None:
                    v = cache[key] = L.sca_gemm_splitk_heuristic(*key) * key[0] * key[1]
                need = max(need, v)
        ws["slab"] = torch.empty(need, dtype=f32, device=dev)
        ws["slab_aux"] = torch.empty(need, dtype=f32, device=dev)
        for ly in self.enc + self.dec:
            ws[ly.tag + ".da"] = torch.empty(B, ly.n_out, dtype=f32, device=dev)
        Dm = Z if (self.mmd_on == "z" or xp) else self.dec[0].n_out
        # The slab / partial counts of the MMD kernels are not monotone in the batch size (fewer column chunks per
        # workgroup at smaller B): size for the worst case over every B' <= B, so that a ragged last minibatch or a
        # validation batch never outgrows buffers whose addresses recorded launch plans hold.  The counts only change
        # at multiples of 16 rows.
        g_need, p_need = 1, 1
        for b in list(range(16, B, 16)) + [B]:
            g_need = max(g_need, L.sca_mmd_num_slabs(b, Dm) * b * Dm)
            p_need = max(p_need, L.sca_mmd_num_partials(b, Dm))
        ws["mmd_g"] = torch.empty(g_need, dtype=f32, device=dev)
        ws["mmd_part"] = torch.zeros(p_need, dtype=torch.float64, device=dev)
        ws["mmd_g1"] = torch.empty(B, Dm, dtype=f32, device=dev)      # the summed gradient the backward pass reads
        self._sbh = torch.empty(B * Hmax, dtype=f32, device=dev)
        ws["Wt"] = torch.empty(C * C, dtype=f32, device=dev)
        ws["counts"] = torch.empty(C, dtype=torch.int32, device=dev)
        self.ws = ws
        self.Bcap = B

    # ------------------------------------------------------------------------------------------ helpers
    def _next_event(self):
        ev = self._events[self._ev_i]
        self._ev_i = (self._ev_i + 1) % len(self._events)
        return ev

    def _fork(self):
        """aux stream waits for everything enqueued on the main stream so far."""
        if not self.overlap:
            return torch.cuda.current_stream(self.device).cuda_stream
        ev = self._next_event()
        self._call(ev.record, torch.cuda.current_stream(self.device))
        self._call(self.aux.wait_event, ev)
        return self.aux.cuda_stream

    def _join(self):
        """main stream waits for everything enqueued on the aux stream so far."""
        if not self.overlap:
            return
        ev = self._next_event()
        self._call(ev.record, self.aux)
        self._call(torch.cuda.current_stream(self.device).wait_event, ev)

    def _call(self, fn, *args):
        """Run a host-side stream operation now and, while a launch plan is being recorded, remember it."""
        if self._rec is not None:
            self._rec.append((fn, args, False))
        return fn(*args)

    def _defer(self, fn, lane=0, join=(), reuse=False):
        """Side work for an aux stream (lane 0: self.aux, lane 1: self.aux2) that depends on everything enqueued on
        the main stream so far.  The dependency point (an event) is taken now, the launches are issued at the next
        _flush(): the host keeps feeding the critical main-stream chain first, the aux streams fill in behind it.
        `join`: lanes whose work queued BEFORE this entry must also have finished (e.g. the gradient tail on lane 0
        sums the slabs of a GEMM that ran on lane 1)."""
        if not self.overlap:
            fn(torch.cuda.current_stream(self.device).cuda_stream)
            return
        if reuse and self._aux_q:
            # same dependency point as the previous deferral (nothing was launched on the main stream in between):
            # share its event - every event recorded between two kernels of the main chain costs ~6 us there
            ev = self._aux_q[-1][0]
        else:
            ev = self._next_event()
            self._call(ev.record, torch.cuda.current_stream(self.device))
        self._aux_q.append((ev, fn, lane, tuple(join)))

    def _lane_stream(self, lane):
        return self.aux if lane == 0 else self.aux2

    def _flush(self):
        """Issue the deferred aux-stream work; returns the events marking its end on the aux streams used so far in
        this step (None without overlap)."""
        if not self.overlap:
            return None
        for ev, fn, lane, join in self._aux_q:
            stream = self._lane_stream(lane)
            self._call(stream.wait_event, ev)
            for other in join:
                if other != lane and other in self._lanes_used:
                    ev2 = self._next_event()
                    self._call(ev2.record, self._lane_stream(other))
                    if (lane, other) in self._side_deps and torch.cuda.is_current_stream_capturing():
                        # ROCm 7.2: hipStreamEndCapture segfaults when two FORKED streams of one capture wait for each
                        # other in both directions (A waits for an event of B after B waited for one of A - acyclic and
                        # legal; torch-only reproduction in scripts/dbg/capture_crash_repro.py, also through a third
                        # stream, also without work in between).  The second direction goes through the origin stream.
                        main = torch.cuda.current_stream(self.device)
                        ev3 = self._next_event()
                        self._call(main.wait_event, ev2)
                        self._call(ev3.record, main)
                        self._call(stream.wait_event, ev3)
                    else:
                        self._call(stream.wait_event, ev2)
                    self._side_deps.add((other, lane))
            self._lanes_used.add(lane)
            fn(stream.cuda_stream)
        self._aux_q = []
        evs = []
        for lane in sorted(self._lanes_used):
            ev = self._next_event()
            self._call(ev.record, self._lane_stream(lane))
            evs.append(ev)
        return tuple(evs)

    def _wait(self, evs):
        """main stream waits for the aux-stream events returned by _flush()."""
        if evs is not None:
            for ev in evs:
                self._call(torch.cuda.current_stream(self.device).wait_event, ev)

    def _zero(self, t, st):
        _lib.check(self.lib.sca_fill_zero(t.data_ptr(), t.numel() * t.element_size(), st), "fill_zero")

    def _gemm(self, am, bm, A, lda, Bm, ldb, M, N, K, st, out=None, ldc=None, bias=None, force_s1=False,
              slab_key="slab", max_split=None):
        """Launch one GEMM; returns (ptr, S, slab_stride).  S == 1 with `out` writes the final tensor directly.
        `max_split` caps the split-K count (the fused middle kernels sum the slabs per 16-row tile)."""
        L = self.lib
        S = 1 if force_s1 else L.sca_gemm_splitk_heuristic(M, N, K)
        if max_split is not None:
            S = min(S, max_split)
        rec = getattr(self, "record_gemms", None)
        if S == 1 and out is not None:
            args = (am, bm, A, lda, Bm, ldb, out, ldc or N, M, N, K, 1, 0, bias)
            if rec is not None:
                rec.append(args)
            _lib.check(L.sca_gemm_f32(*args, st), "gemm")
            return out, 1, 0
        if S * M * N > self.ws[slab_key].numel():
            raise RuntimeError("split-K slabs of a %dx%dx%d GEMM (S=%d) exceed the workspace (%d floats)"
                               % (M, N, K, S, self.ws[slab_key].numel()))
        slab = self.ws[slab_key].data_ptr()
        args = (am, bm, A, lda, Bm, ldb, slab, N, M, N, K, S, M * N, None)
        if rec is not None:
            rec.append(args)
        _lib.check(L.sca_gemm_f32(*args, st), "gemm")
        return slab, S, M * N

    def replay_gemms(self, calls):
        """Re-issue recorded GEMM launches (bench.py roofline leg: same shapes, same buffers, nothing else)."""
        st = torch.cuda.current_stream(self.device).cuda_stream
        for args in calls:
            if args[0] == "kk":
                _lib.check(self.lib.sca_gemm_kk_f32(*args[1], st), "gemm_kk")
            elif args[0] == "group":
                _lib.check(self.lib.sca_gemm_group_f32(args[1], len(args[1]), st), "gemm_group")
            else:
                _lib.check(self.lib.sca_gemm_f32(*args, st), "gemm")

    def _gemm_to(self, am, bm, A, lda, Bm, ldb, M, N, K, st, dst, slab_key="slab"):
        """GEMM whose (reduced) result lands in the contiguous tensor `dst` [M, N]."""
        L = self.lib
        p, S, stride = self._gemm(am, bm, A, lda, Bm, ldb, M, N, K, st, out=dst.data_ptr(), ldc=N, slab_key=slab_key)
        if p != dst.data_ptr():
            _lib.check(L.sca_splitk_reduce(p, S, stride, dst.data_ptr(), M * N, st), "splitk_reduce")

    # ---- gradient tail: slab sums and column reductions that only feed Adam, batched into one launch ------------
    def _grad_gemm(self, am, bm, A, lda, Bm, ldb, M, N, K, st, dst, allow_kk=True):
        """Weight-gradient GEMM into `dst` [M, N]: split-K slabs go to a region of their own and their sum becomes a
        job of the tail kernel (instead of a sca_splitk_reduce launch per tensor)."""
        L = self.lib
        rec = getattr(self, "record_gemms", None)
        # large dY^T X products: the register-direct k-major kernel (gemm_f32.hip, gemm_kk_kernel), one launch each
        if (allow_kk and am == 1 and bm == 1 and M * N >= self.KK_MIN_MN and K >= 1024
                and L.sca_gemm_kk_supported(A, lda, Bm, ldb, M, N)):
            S = L.sca_gemm_kk_splitk_heuristic(M, N, K)
            if S == 1:
                out, stride = dst.data_ptr(), 0
            else:
                key = "gslab:%d" % dst.data_ptr()
                slab = self.ws.get(key)
                if slab is None or slab.numel() < S * M * N:
                    if slab is not None:
                        self._invalidate()
                    slab = self.ws[key] = torch.empty(S * M * N, dtype=torch.float32, device=self.device)
                out, stride = slab.data_ptr(), M * N
                self._tail_jobs.append((0, S, M * N, out, dst.data_ptr(), M * N, 0, 0, 0, 0, 0))
            args = (A, lda, Bm, ldb, out, N, M, N, K, S, stride)
            if rec is not None:
                rec.append(("kk", args))
            _lib.check(L.sca_gemm_kk_f32(*args, st), "gemm_kk")
            return
        S = L.sca_gemm_splitk_heuristic(M, N, K)
        if S == 1:
            args = (am, bm, A, lda, Bm, ldb, dst.data_ptr(), N, M, N, K, 1, 0, None)
            if rec is not None:
                rec.append(args)
            _lib.check(L.sca_gemm_f32(*args, st), "gemm")
            return
        key = "gslab:%d" % dst.data_ptr()
        slab = self.ws.get(key)
        if slab is None or slab.numel() < S * M * N:
            if slab is not None:
                self._invalidate()      # recorded plans hold the old address
            slab = self.ws[key] = torch.empty(S * M * N, dtype=torch.float32, device=self.device)
        args = (am, bm, A, lda, Bm, ldb, slab.data_ptr(), N, M, N, K, S, M * N, None)
        self._tail_jobs.append((0, S, M * N, slab.data_ptr(), dst.data_ptr(), M * N, 0, 0, 0, 0, 0))
        group = getattr(self, "_gg_jobs", None)
        if group is not None and am == 1 and bm == 1:      # joins the grouped launch (_launch_group)
            group.append(args)
            return
        if rec is not None:
            rec.append(args)
        _lib.check(L.sca_gemm_f32(*args, st), "gemm")

    def _launch_group(self, jobs, st):
        """Weight-gradient GEMMs collected by _grad_gemm -> sca_gemm_group_f32 launches (one for up to 8 jobs)."""
        if not jobs:
            return
        rec = getattr(self, "record_gemms", None)
        cache = self.__dict__.setdefault("_gg_cache", {})
        key = tuple(jobs)
        arrs = cache.get(key)
        if arrs is None:
            if len(cache) > 64:
                cache.clear()
            mx = self.lib.sca_gemm_group_max_jobs()
            arrs = []
            for lo in range(0, len(jobs), mx):
                chunk = jobs[lo:lo + mx]
                arr = (_lib.GemmJob * len(chunk))()
                for j, (am, bm, A, lda, Bm, ldb, Cp, ldc, M, N, K, S, stride, _) in zip(arr, chunk):
                    j.amode, j.bmode, j.A, j.B, j.C = am, bm, A, Bm, Cp
                    j.lda, j.ldb, j.ldc, j.M, j.N, j.K, j.splitk, j.slab_stride = lda, ldb, ldc, M, N, K, S, stride
                arrs.append((arr, tuple(chunk)))
            cache[key] = arrs
        for arr, chunk in arrs:
            if rec is not None:
                rec.append(("group", arr, chunk))
            _lib.check(self.lib.sca_gemm_group_f32(arr, len(arr), st), "gemm_group")

    def _colsum(self, da, B, H, cond_p, C, dWc, db, st):
        """db[h] = sum_b da[b][h]; dWc[h][c] = sum over rows of condition c (tail job; direct launch for huge C)."""
        if dWc is None and db is None:
            return
        if dWc is not None and C > 512:
            _lib.check(self.lib.sca_cond_colsum(da, B, H, cond_p, C, self.ws["segT"].data_ptr(), dWc, db, st),
                       "cond_colsum")
            return
        self._tail_jobs.append((1, 0, 0, da, db or 0, 0, dWc or 0, (cond_p or 0) if dWc else 0, B, H,
                                C if dWc else 0))

    def _defer_tail(self, join=(), lane=0):
        """Gradient tail on an aux stream and - single-device train_step - Adam right behind it on the same stream (no
        cross-stream hop between the last reduction and the update)."""
        self._defer(self._run_tail, lane=lane, join=join)
        fused = self._adam_after_tail
        if fused is not None:
            if self._dp_split:
                # data parallel: the ONE all-reduce of the flat gradient buffer is part of the step's recorded launch
                # sequence, issued from the stream the tail runs on (RCCL orders itself behind the stream it is called
                # from), so it starts the moment the last reduction is done and Adam follows it on the same stream
                self._defer(lambda sa: self._call(self._dp_allreduce_on, lane), lane=lane)
            self._defer(lambda sa: self.adam_step(fused[0], fused[1], stream=sa), lane=lane)
            self._adam_done = True

    def _dp_allreduce_on(self, lane):
        """Gradient all-reduce of a data-parallel step (a host call recorded into the launch plan).  Warm-up and
        recording passes skip it: a rank records a plan when ITS batch size is new, so the collective would be
        unmatched on the other ranks (_dp_dry)."""
        if self._dp_dry or self._dp_world <= 1:
            return
        from . import distributed as sdist
        with torch.cuda.stream(self._lane_stream(lane)):
            sdist.allreduce_mean_(self.flat_g, self._dp_world, weight=self._dp_grad_weight)

    def _run_tail(self, st):
        jobs, self._tail_jobs = self._tail_jobs, []
        if not jobs:
            return
        cache = self.__dict__.setdefault("_tail_cache", {})
        key = tuple(jobs)
        arrs = cache.get(key)
        if arrs is None:
            if len(cache) > 64:
                cache.clear()
            mx = self.lib.sca_grad_tail_max_jobs()
            arrs = []
            for lo in range(0, len(jobs), mx):
                chunk = jobs[lo:lo + mx]
                arr = (_lib.TailJob * len(chunk))()
                for j, (kind, S, stride, src, dst, n, dWc, cond, B, H, C) in zip(arr, chunk):
                    j.kind, j.S, j.stride, j.src, j.dst, j.n = kind, S, stride, src, dst or None, n
                    j.dWc, j.cond, j.B, j.H, j.C = dWc or None, cond or None, B, H, C
                arrs.append(arr)
            cache[key] = arrs
        for arr in arrs:
            _lib.check(self.lib.sca_grad_tail(arr, len(arr), st), "grad_tail")

    def _layer_block_fwd(self, ly, sp, S, stride, B, cond_args, p_drop, mask_ptr, train, step_p, st):
        """Everything between two GEMMs of one layer: slab sum + bias + condition term, then LayerNorm or BatchNorm,
        ReLU, dropout.  Leaves the normalised pre-activation in ws[tag.yn] and the output in ws[tag.h]."""
        L, P, ws = self.lib, self.p, self.ws
        ptr = _lib.ptr
        cw, cwp, cvec_p, cE = cond_args
        yn, h = ws[ly.tag + ".yn"].data_ptr(), ws[ly.tag + ".h"].data_ptr()
        if ly.bn is None:
            _lib.check(L.sca_layer_fwd(sp, S, stride, B, ly.n_out, ptr(P[ly.b]) if ly.b else None,
                                       ptr(P[ly.wc]) if ly.wc else None, cw, cwp, cvec_p, cE, self.norm, 1, p_drop,
                                       mask_ptr, self.seed, ly.rng_stream, step_p, yn,
                                       ws[ly.tag + ".rs"].data_ptr(), h, st), "layer_fwd")
            return h
        # BatchNorm: pre-activation first (row kernel without norm / activation), then the batch-wise part
        _lib.check(L.sca_layer_fwd(sp, S, stride, B, ly.n_out, ptr(P[ly.b]) if ly.b else None,
                                   ptr(P[ly.wc]) if ly.wc else None, cw, cwp, cvec_p, cE, 0, 0, 0.0, None,
                                   self.seed, ly.rng_stream, None, yn, None, yn, st), "layer_pre")
        bw = ly.bn + ".weight"
        # frozen surgery models stop tracking running statistics (trainers/trvae/trainer.py:264-270)
        update = bool(train) and not (self.freeze and not self.trainable[bw])
        _lib.check(L.sca_bn_fwd(yn, B, ly.n_out, P[bw].data_ptr(), P[ly.bn + ".bias"].data_ptr(),
                                self.buffers[ly.bn + ".running_mean"].data_ptr(),
                                self.buffers[ly.bn + ".running_var"].data_ptr(), int(bool(train)), int(update), 0.1,
                                p_drop, mask_ptr, self.seed, ly.rng_stream, step_p, ws["bn_acc"].data_ptr(),
                                ws[ly.tag + ".bnstats"].data_ptr(), h, st), "bn_fwd")
        if update:
            self._call(self.buffers[ly.bn + ".num_batches_tracked"].add_, 1)
        return h

    # ------------------------------------------------------------------------------------------ the step
    def forward_backward(self, row_ids: Optional[torch.Tensor], cond: Optional[torch.Tensor], B: int, train=True,
                         compute_grad=True, eps: Optional[torch.Tensor] = None,
                         masks: Optional[Dict[str, torch.Tensor]] = None, accumulate_logs=True,
                         x_dense: Optional[torch.Tensor] = None, covs: Optional[torch.Tensor] = None,
                         proto: Optional[dict] = None, n_global: Optional[int] = None):
        """One fwd(+bwd) pass over the minibatch `row_ids` (int64 device tensor of CSR row numbers) whose conditions
        are `cond` (int32 device tensor).  Leaves losses in self.losses = [recon, kl, mmd, total] and, when
        compute_grad, the gradients of recon + hyper[0]*kl + mmd in self.flat_g.  Nothing is synchronised."""
        L, P = self.lib, self.p
        self._ensure(B)
        eps = self._eps_or_zero(eps, B)
        ws = self.ws
        st = torch.cuda.current_stream(self.device).cuda_stream
        G, C, Z = self.G, self.C, self.Zt
        ptr = _lib.ptr
        p_drop = self.dr_rate if train else 0.0
        cond_p = ptr(cond) if C > 0 else None
        step_p = self.step_ctr.data_ptr()
        masks = masks or {}
        self._p_drop_cur, self._masks_cur = p_drop, masks
        self._cond_cur = cond
        if n_global is not None:        # data-parallel runs: size of the GLOBAL minibatch this slice belongs to
            self._dp_n_global = int(n_global)

        # 1. minibatch gather (+ log1p, size factors)
        log1p = 0 if self.recon == "mse" else 1
        if x_dense is not None:
            _lib.check(L.sca_dense_gather_rows(ptr(x_dense), x_dense.stride(0), ptr(row_ids), B, G, log1p,
                                               ws["x"].data_ptr(), G, ws["sf"].data_ptr(), st), "dense_gather")
            self._pf = None
        else:
            ip, ix, dv = self.csr
            if not self._pf_flags[0]:       # (True: the previous step already gathered this minibatch, train_step)
                _lib.check(L.sca_csr_gather_rows(ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), ptr(row_ids), B, G, log1p,
                                                 ws["x"].data_ptr(), G, ws["sf"].data_ptr(), st), "csr_gather")
                self._pf = None

        self._aux_q = []
        self._lanes_used = set()
        self._side_deps = set()
        self._tail_jobs = []
        self._prologue(B, compute_grad)
        self._covs_cur = covs
        if self.cond_embed:
            self._embed_lookup(covs, B, st)
            cw, cwp, cvec_p, cE = 0, None, ws["cvec"].data_ptr(), self.E
        else:
            cw, cwp, cvec_p, cE = C, cond_p, None, 0

        cond_args = (cw, cwp, cvec_p, cE)

        def layer_forward(ly, a_in):
            sp, S, stride = self._gemm(0, 0, a_in, ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out, ly.n_in, st)
            return self._layer_block_fwd(ly, sp, S, stride, B, cond_args, p_drop, ptr(masks.get(ly.mask_key)), train,
                                         step_p, st)

        # 2. encoder
        fused = self.use_fused_mid and not (self.use_mmd and self.mmd_on != "z")
        a = ws["x"].data_ptr()
        if fused:       # first layer: GEMM only, its epilogue runs inside mid_fwd
            ly = self.enc[0]
            pre0 = self._gemm(0, 0, a, ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out, ly.n_in, st, max_split=self._mid_max_split)
            self._ev_pro = self._flush()
            return self._fused_path(row_ids, cond_p, B, train, compute_grad, eps, masks, accumulate_logs, p_drop, st,
                                    pre0)
        for ly in self.enc:
            a = layer_forward(ly, a)
        self._ev_pro = self._flush()
        # 3. heads + sampling + KL
        Hl = self.H[-1]
        # heads: one [2*Zt, H] GEMM, bias fused, no split-K (K = H is short) -> the latent kernel reads one slab
        self._gemm(0, 0, a, Hl, self.heads_w.data_ptr(), Hl, B, 2 * Z, Hl, st, out=ws["mu_lv"].data_ptr(),
                   ldc=2 * Z, bias=self.heads_b.data_ptr(), force_s1=True)
        _lib.check(L.sca_latent_fwd(ws["mu_lv"].data_ptr(), 1, 0, None, B, Z, ptr(eps), self.seed, 1, step_p, 1,
                                    ws["mu_lv"].data_ptr(), ws["eps"].data_ptr(), ws["z"].data_ptr(),
                                    ws["kl_row"].data_ptr(), st), "latent_fwd")
        # Side work is deferred: its dependency point is taken here, its launches are issued after the host has fed
        # the main-stream chain (the step is launch-rate sensitive; the critical path must never wait for the host).
        do_mmd = self.use_mmd and C > 0
        self._n_part = 0
        self._dz2_cur = (None, 0)
        self._mmd_ev = None

        def mmd_work(sa):
            feat, Dm = (ws["z"], Z) if self.mmd_on == "z" else (ws["dec0.h"], self.dec[0].n_out)
            self._mmd_launch(sa, feat, Dm, cond, cond_p, B, compute_grad)

        def want_mmd():
            Dm = Z if self.mmd_on == "z" else self.dec[0].n_out
            self._n_part = self._mmd_partials(B, Dm)
            self._defer(mmd_work)

        if do_mmd and self.mmd_on == "z":
            want_mmd()
        if self.cond_embed and proto is not None:
            self._n_part = self._launch_proto(proto, B, compute_grad)
        ev_z = self._flush()         # MMD / prototype losses start now, beside the (small) decoder kernels
        if self.arch == "expimap":
            return self._expimap_tail(row_ids, cond_p, B, compute_grad, accumulate_logs, st)
        # 4. decoder
        a = ws["z"].data_ptr()
        for i, ly in enumerate(self.dec):
            a = layer_forward(ly, a)
            if i == 0 and do_mmd and self.mmd_on != "z":
                want_mmd()
        Hd = self.dec[-1].n_out
        # 5. output layer(s); the aux stream meanwhile prepares the theta tables / zeroes the accumulating gradient
        #    targets and runs the MMD / prototype losses
        self._gemm(0, 0, a, Hd, P[self.out_w].data_ptr(), Hd, B, G, Hd, st, out=ws["logits"].data_ptr(), ldc=G,
                   bias=P[self.out_b].data_ptr(), force_s1=True)
        zinb = self.recon == "zinb"
        if zinb:
            self._gemm(0, 0, a, Hd, P["decoder.dropout_decoder.weight"].data_ptr(), Hd, B, G, Hd, st,
                       out=ws["pi"].data_ptr(), ldc=G, bias=P["decoder.dropout_decoder.bias"].data_ptr(),
                       force_s1=True)
        ev_side = self._flush() if self._aux_q else ev_z
        self._wait(self._ev_tables or self._ev_pro)
        # 6. reconstruction loss (+ gradient w.r.t. the logits, in place)
        rpb = _recon_rows_per_block(B)
        gob = self.g[self.out_b].data_ptr() if self.trainable[self.out_b] else None
        if self.recon == "mse":
            _lib.check(L.sca_mse_rows(ws["logits"].data_ptr(), G, ws["x"].data_ptr(), G, B, G, rpb,
                                      ws["row_ll"].data_ptr(), gob, int(compute_grad), 1, None, None, st), "mse_rows")
            recon_sign = 1.0
        elif self.recon == "poisson":
            ip, ix, dv = self.csr
            _lib.check(L.sca_poisson_rows(ws["logits"].data_ptr(), G, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(),
                                          ptr(row_ids), ws["sf"].data_ptr(), B, G, rpb, ws["row_ll"].data_ptr(),
                                          int(compute_grad), st), "poisson_rows")
            recon_sign = 1.0
        else:
            ip, ix, dv = self.csr
            _lib.check(L.sca_recon_rows_tab(int(zinb), ws["logits"].data_ptr(), G,
                                            ws["pi"].data_ptr() if zinb else None,
                                            ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), ptr(row_ids),
                                            ws["sf"].data_ptr(), cond_p, ws["tab_theta"].data_ptr(),
                                            ws["tab_lte"].data_ptr(), B, G, max(C, 1), rpb, ws["row_ll"].data_ptr(),
                                            ws["dthetaT"].data_ptr(), gob,
                                            self.g["decoder.dropout_decoder.bias"].data_ptr()
                                            if zinb and self.trainable["decoder.dropout_decoder.bias"] else None,
                                            int(compute_grad), None, None, ptr(ws.get("gam_tab")), st), "recon_rows")
            recon_sign = -1.0
        self._defer_finalize(recon_sign, B, accumulate_logs)
        if not compute_grad:
            self._wait(self._flush())
            return self.losses
        self._backward(B, cond_p, p_drop, masks, st, do_mmd, ev_side)
        return self.losses

    def _prologue(self, B, compute_grad):
        """Start-of-step side work (aux stream, deferred): exp(theta) tables for the likelihood kernel and zeroing of
        the gradient targets that kernels accumulate into with atomics (every other gradient is overwritten)."""
        L, P, ws = self.lib, self.p, self.ws
        G, C = self.G, max(self.C, 1)
        xp_c = self.arch == "expimap" and self.xp_wc

        def work(sa):
            # theta tables + dthetaT = 0 + optimizer.zero_grad() (most gradients are overwritten, the accumulating ones -
            # output bias, column sums of large batches, embedding scatter - need zeros: one fill of the flat buffer
            # covers all) in ONE launch where the buffers allow it
            zt = ws["dthetaT"] if (self.has_theta and compute_grad) else None
            zg = self.flat_g if compute_grad else None
            nbytes = lambda t: t.numel() * t.element_size()
            fused_ok = self.has_theta and "gam_tab" not in ws and all(
                t is None or (t.data_ptr() % 16 == 0 and nbytes(t) % 16 == 0) for t in (zt, zg))
            if fused_ok:
                _lib.check(L.sca_theta_prep_zero(P["theta"].data_ptr(), G, C, ws["tab_theta"].data_ptr(),
                                                 ws["tab_lte"].data_ptr(), _lib.ptr(zt), nbytes(zt) if zt is not None else 0,
                                                 _lib.ptr(zg), nbytes(zg) if zg is not None else 0, sa), "theta_prep_zero")
            else:
                if self.has_theta:
                    _lib.check(L.sca_theta_prep(P["theta"].data_ptr(), G, C, ws["tab_theta"].data_ptr(),
                                                ws["tab_lte"].data_ptr(), sa), "theta_prep")
                    if "gam_tab" in ws:
                        _lib.check(L.sca_nb_gamma_table(ws["tab_theta"].data_ptr(), C, G, ws["gam_tab"].data_ptr(), sa),
                                   "nb_gamma_table")
                if zt is not None:
                    self._zero(zt, sa)
                if zg is not None:
                    self._zero(zg, sa)
            if xp_c:
                _lib.check(L.sca_transpose_scale(P[self.xp_wc].data_ptr(), self.G, self.C, 1.0,
                                                 ws["coffT"].data_ptr(), sa), "cond_transpose")
            if self._mmd_prepared:
                if self.overlap:
                    # what the likelihood kernel waits for ends here: the MMD's weight table below is only read by the
                    # MMD launch, which follows it on this stream (at 128 cells the table kernel ended 12 us after the
                    # encoder; the output GEMM / likelihood kernel waited for it for nothing)
                    ev = self._next_event()
                    self._call(ev.record, self.aux)
                    self._ev_tables = (ev,)
                # the MMD weight table only needs the conditions: built here, a whole encoder ahead of the MMD launch
                # (a single latency-bound workgroup, ~25 us at 8192 cells, that used to sit in front of the MMD kernel)
                bnd = -1 if self.boundary is None else int(self.boundary)
                _lib.check(L.sca_mmd_prepare(_lib.ptr(self._cond_cur), B, self.C, bnd, self.beta, ws["Wt"].data_ptr(),
                                             ws["counts"].data_ptr(), sa), "mmd_prepare")
        Dm = self.Zt if (self.mmd_on == "z" or self.arch == "expimap") else self.dec[0].n_out
        self._mmd_prepared = bool(self.use_mmd and self.C > 0 and not self._dp_exact() and self._cond_cur is not None
                                  and L.sca_mmd_prepared_supported(B, Dm))
        self._ev_tables = None
        self._defer(work)

    def _defer_finalize(self, recon_sign, B, accumulate_logs):
        """Loss composition / epoch-log accumulation (aux stream: nothing on the critical path reads it)."""
        L, ws = self.lib, self.ws
        n_part = self._n_part

        def work(sa):
            _lib.check(L.sca_finalize_losses(ws["row_ll"].data_ptr(), recon_sign, ws["kl_row"].data_ptr(), B,
                                             ws["mmd_part"].data_ptr(), n_part, self.hyper.data_ptr(),
                                             self.losses.data_ptr(),
                                             self.accum.data_ptr() if accumulate_logs else None, sa), "finalize")
        self._defer(work)

    def _backward(self, B, cond_p, p_drop, masks, st, have_mmd, ev_side):
        """Backward pass.  The dX chain (layer_bwd -> dX GEMM -> ...) is the critical path and stays on the main
        stream; every weight / bias / condition-weight gradient only feeds Adam, so it is deferred to the aux stream
        (own slab workspace, per-layer `da` buffers) and joined once at the end."""
        L, P, Gd, ws, T = self.lib, self.p, self.g, self.ws, self.trainable
        G, C, Z = self.G, self.C, self.Zt
        AUX = "slab_aux" if self.overlap else "slab"
        Hd = self.dec[-1].n_out
        h_last = ws[self.dec[-1].tag + ".h"].data_ptr()
        dl = ws["logits"].data_ptr()

        # ---- side work that depends only on the reconstruction kernel's outputs
        def recon_side(sa):
            if self.has_theta and T["theta"]:
                # theta gradient: -1/B * sum_b theta dll/dtheta, transposed back to the [G][C] parameter layout
                _lib.check(L.sca_transpose_scale(ws["dthetaT"].data_ptr(), max(C, 1), G, -1.0 / B,
                                                 Gd["theta"].data_ptr(), sa), "transpose_scale")
            if T[self.out_w]:
                self._grad_gemm(1, 1, dl, G, h_last, Hd, G, Hd, B, sa, Gd[self.out_w])
            if self.recon == "poisson" and T[self.out_b]:      # (the other likelihood kernels accumulate it themselves)
                self._colsum(dl, B, G, None, 0, None, Gd[self.out_b].data_ptr(), sa)
            if self.recon == "zinb" and T["decoder.dropout_decoder.weight"]:
                self._grad_gemm(1, 1, ws["pi"].data_ptr(), G, h_last, Hd, G, Hd, B, sa,
                                Gd["decoder.dropout_decoder.weight"])
        self._defer(recon_side)
        extra = None
        if self.recon == "zinb":
            # dh = dl @ Wo + dpi @ Wp : the second product is reduced into a scratch [B, Hd] and joins as `extra`
            scratch = self._scratch_bh(B, Hd)
            self._gemm_to(0, 1, ws["pi"].data_ptr(), G, P["decoder.dropout_decoder.weight"].data_ptr(), Hd, B, Hd, G,
                          st, scratch)
            extra = scratch.data_ptr()
        sp, S, stride = self._gemm(0, 1, dl, G, P[self.out_w].data_ptr(), Hd, B, Hd, G, st)

        # ---- decoder, last -> first
        for li in range(len(self.dec) - 1, -1, -1):
            ly = self.dec[li]
            if li == 0 and have_mmd and self.mmd_on == "y":
                # gradient of the MMD on y1 joins the upstream gradient of the first decoder layer
                self._wait(ev_side)
                if extra is not None:       # single-hidden-layer ZINB net: both extras land on the same layer
                    n = B * ly.n_out
                    self._call(ws["mmd_g1"].view(-1)[:n].add_, self._scratch_bh(B, ly.n_out).view(-1))
                extra = ws["mmd_g1"].data_ptr()
            a_prev = ws["z"].data_ptr() if li == 0 else ws[self.dec[li - 1].tag + ".h"].data_ptr()
            sp, S, stride = self._layer_backward(ly, a_prev, sp, S, stride, extra, True, B, cond_p, p_drop, masks, st)
            extra = None

        if have_mmd and self.mmd_on == "z":
            dz2, S2 = ws["mmd_g1"].data_ptr(), 1
        else:
            dz2, S2 = self._dz2_cur
        if S2:
            self._wait(ev_side)         # MMD / prototype gradient w.r.t. z (finished long ago on the aux stream)
        self._backward_latent_encoder(B, cond_p, p_drop, masks, st, sp, S, stride, dz2, S2)

    def _layer_backward(self, ly, a_prev, sp, S, stride, extra, need_dx, B, cond_p, p_drop, masks, st,
                        have_da=False, side_out=None):
        """Gradient through one Linear(+cond) -> norm -> ReLU -> dropout block: `da` and (need_dx) the dX GEMM slabs for
        the layer below on the main stream; the weight / bias / condition-weight gradients are deferred to the aux
        stream."""
        L, P, Gd, ws, T = self.lib, self.p, self.g, self.ws, self.trainable
        C = self.C
        ptr = _lib.ptr
        AUX = "slab_aux" if self.overlap else "slab"
        m = masks.get(ly.mask_key)
        da = ws[ly.tag + ".da"].data_ptr()
        if have_da:
            pass
        elif ly.bn is None:
            _lib.check(L.sca_layer_bwd(sp, S, stride, extra, B, ly.n_out, ws[ly.tag + ".yn"].data_ptr(),
                                       ws[ly.tag + ".rs"].data_ptr(), self.norm, 1, p_drop, ptr(m), self.seed,
                                       ly.rng_stream, self.step_ctr.data_ptr(), da, st), "layer_bwd")
        else:
            bw, bb = ly.bn + ".weight", ly.bn + ".bias"
            _lib.check(L.sca_bn_bwd(sp, S, stride, extra, B, ly.n_out, ws[ly.tag + ".yn"].data_ptr(),
                                    ws[ly.tag + ".bnstats"].data_ptr(), P[bw].data_ptr(), P[bb].data_ptr(), 1, p_drop,
                                    ptr(m), self.seed, ly.rng_stream, self.step_ctr.data_ptr(),
                                    ws["bn_acc"].data_ptr(), da, Gd[bw].data_ptr() if T[bw] else None,
                                    Gd[bb].data_ptr() if T[bb] else None, st), "bn_bwd")
        need_w = T[ly.w]
        need_b = bool(ly.b) and T[ly.b]
        need_c = bool(ly.wc) and T[ly.wc]
        need_e = self.cond_embed and bool(ly.wc) and any(T[k] for (k, _, _) in self.emb)

        def side(sa):
            if need_w:
                self._grad_gemm(1, 1, da, ly.n_out, a_prev, ly.n_in, ly.n_out, ly.n_in, B, sa, Gd[ly.w])
            if self.cond_embed:
                if need_c:       # dWc [H, E] = da^T @ cvec
                    self._grad_gemm(1, 1, da, ly.n_out, ws["cvec"].data_ptr(), self.E, ly.n_out, self.E, B, sa,
                                    Gd[ly.wc])
                if need_b:
                    self._colsum(da, B, ly.n_out, None, 0, None, Gd[ly.b].data_ptr(), sa)
                if need_e:
                    # gradient w.r.t. the dense condition vector: dcvec [B, E] = da @ Wc (scattered into the
                    # embedding tables at the end of the backward pass)
                    self._gemm_to(0, 1, da, ly.n_out, P[ly.wc].data_ptr(), self.E, B, self.E, ly.n_out, sa,
                                  ws["dcvec_enc" if ly.tag.startswith("enc") else "dcvec_dec"], slab_key=AUX)
            elif need_b or need_c:
                self._colsum(da, B, ly.n_out, cond_p, C, Gd[ly.wc].data_ptr() if need_c else None,
                             Gd[ly.b].data_ptr() if need_b else None, sa)
        if need_w or need_b or need_c or need_e:
            if side_out is not None:
                side_out.append(side)
            else:
                self._defer(side)
        if need_dx:
            return self._gemm(0, 1, da, ly.n_out, P[ly.w].data_ptr(), ly.n_in, B, ly.n_in, ly.n_out, st)
        return None, 0, 0

    def _backward_latent_encoder(self, B, cond_p, p_drop, masks, st, sp, S, stride, dz2, S2):
        """Latent block (dz from the decoder + dz2 from MMD / HSIC / prototypes + KL) -> heads -> encoder layers."""
        L, ws, T = self.lib, self.ws, self.trainable
        Z = self.Zt
        AUX = "slab_aux" if self.overlap else "slab"
        _lib.check(L.sca_latent_bwd(sp, S, stride, dz2, S2, B * Z, ws["mu_lv"].data_ptr(),
                                    ws["eps"].data_ptr(), self.hyper.data_ptr(), B, Z, ws["dheads"].data_ptr(), st),
                   "latent_bwd")
        Hl = self.H[-1]
        h_enc = ws[self.enc[-1].tag + ".h"].data_ptr()
        dh = ws["dheads"].data_ptr()
        if any(T[k] for k in self.heads_names):
            # mean / log_var (/ expand) heads share one fused gradient GEMM; a frozen member's slice of the result is
            # simply never consumed by Adam (trainable_ranges() skips it)
            def heads_side(sa):
                self._grad_gemm(1, 1, dh, 2 * Z, h_enc, Hl, 2 * Z, Hl, B, sa, self.heads_gw)
                self._colsum(dh, B, 2 * Z, None, 0, None, self.heads_gb.data_ptr(), sa)
            self._defer(heads_side)
        sp, S, stride = self._gemm(0, 1, dh, 2 * Z, self.heads_w.data_ptr(), Hl, B, Hl, 2 * Z, st)
        for li in range(len(self.enc) - 1, -1, -1):
            ly = self.enc[li]
            a_prev = ws["x"].data_ptr() if li == 0 else ws[self.enc[li - 1].tag + ".h"].data_ptr()
            sp, S, stride = self._layer_backward(ly, a_prev, sp, S, stride, None, li > 0, B, cond_p, p_drop, masks,
                                                 st)
        if self.cond_embed and any(T[k] for (k, _, _) in self.emb):
            self._defer(lambda sa: self._embed_scatter(B, sa))
        self._defer_tail()
        self._wait(self._flush())

    # ------------------------------------------------------------------------------------------ fused middle
    def _mid_args(self, B, cond_p, p_drop, eps, masks, dh=None, S=0, stride=0, dz2=None, pre0=None, sample=1,
                  encoder_only=False):
        """sca_mid_args for the current workspace (cached: every pointer is static for a given batch capacity).
        encoder_only: n_dec = 0 - the forward kernel stops behind the latent block (get_latent); the decoder entries
        stay filled, the straight-line kernels prefetch the first decoder layer's operands before they return."""
        P, ws, ptr = self.p, self.ws, _lib.ptr
        key = (B, self._ws_gen, self.seed, cond_p, p_drop, ptr(eps), tuple(sorted((k, v.data_ptr()) for k, v in masks.items())), dh, S,
               stride, dz2, pre0, sample, encoder_only)
        cache = self.__dict__.setdefault("_mid_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit[1]
        if len(cache) > 32:
            cache.clear()
        a = _lib.MidArgs()

        def fill(dst, ly):
            dst.W, dst.bias = P[ly.w].data_ptr(), ptr(P[ly.b]) if ly.b else None
            dst.Wc = ptr(P[ly.wc]) if ly.wc else None
            dst.mask = ptr(masks.get(ly.mask_key))
            dst.yn, dst.rs = ws[ly.tag + ".yn"].data_ptr(), ws[ly.tag + ".rs"].data_ptr()
            dst.h, dst.da = ws[ly.tag + ".h"].data_ptr(), ws[ly.tag + ".da"].data_ptr()
            if encoder_only:        # eval-mode encoder pass: nothing is kept for a backward pass
                dst.yn = dst.rs = dst.h = None
            dst.N, dst.K, dst.rng_stream = ly.n_out, ly.n_in, ly.rng_stream
        for i, ly in enumerate(self.enc[1:]):
            fill(a.enc[i], ly)
        for i, ly in enumerate(self.dec):
            fill(a.dec[i], ly)
        a.n_enc, a.n_dec = len(self.enc) - 1, 0 if encoder_only else len(self.dec)
        a.Wh, a.bh, a.eps_in = self.heads_w.data_ptr(), self.heads_b.data_ptr(), ptr(eps)
        a.mu_lv, a.eps_out = ws["mu_lv"].data_ptr(), ws["eps"].data_ptr()
        a.z, a.kl_row = ws["z"].data_ptr(), ws["kl_row"].data_ptr()
        a.Z, a.Kh, a.rng_latent, a.sample = self.Zt, self.H[-1], 1, int(sample)
        a.B, a.C, a.cond = B, self.C, cond_p
        a.norm, a.p, a.seed, a.step = self.norm, p_drop, self.seed, self.step_ctr.data_ptr()
        a.h_in, a.K0, a.S = ws[self.enc[0].tag + ".h"].data_ptr(), self.H[0], S
        a.dh, a.dh_stride, a.dz2 = dh, stride, dz2
        a.hyper, a.dheads, a.dh_out = self.hyper.data_ptr(), ws["dheads"].data_ptr(), ws["dh0"].data_ptr()
        if pre0 is not None:      # first encoder layer's epilogue inside the fused kernels
            a.fuse_first, (a.pre0, a.S0, a.pre0_stride) = 1, pre0
            fill(a.first, self.enc[0])
        import ctypes
        cache[key] = (a, ctypes.byref(a))
        return cache[key][1]

    def _fused_path(self, row_ids, cond_p, B, train, compute_grad, eps, masks, accumulate_logs, p_drop, st, pre0):
        """Step after the first encoder layer with the fused middle kernels: mid_fwd -> (MMD beside) output GEMM ->
        likelihood -> dX of the output GEMM -> mid_bwd -> first encoder layer backward; weight-side gradients on the
        aux stream as in the unfused path."""
        L, P, Gd, ws, T = self.lib, self.p, self.g, self.ws, self.trainable
        G, C, Z = self.G, self.C, self.Zt
        ptr = _lib.ptr
        _lib.check(L.sca_mid_fwd(self._mid_args(B, cond_p, p_drop, eps, masks, pre0=pre0), st), "mid_fwd")
        do_mmd = self.use_mmd and C > 0
        self._n_part = 0
        # The MMD and the likelihood kernel are both VALU-bound, the GEMMs around them MFMA-bound: from MMD_SPLIT_MIN
        # cells on the MMD goes in two row blocks - the first beside the output GEMM (and the head of the likelihood),
        # the second behind the likelihood kernel, beside the dX / dW GEMMs of the output layer - instead of sharing the
        # vector units with the likelihood kernel for its whole length.
        n1 = B
        if (do_mmd and compute_grad and self.overlap and self._mmd_prepared and B >= MMD_SPLIT_MIN
                and 0.0 < MMD_SPLIT_FRAC < 1.0):
            n1 = max(256, (int(B * MMD_SPLIT_FRAC) // 256) * 256)
        mmd_split = n1 < B
        # Where the MMD forks off.  Up to MMD_AFTER_OUT_MAX cells: behind the output GEMM, so that the likelihood kernel
        # (which follows that GEMM directly on the main stream) is placed on the CUs first and the MMD fills what is
        # left - forked earlier, the MMD's workgroups hold the CUs and the likelihood workgroups trickle in behind
        # them (4096 cells: 0.361 -> 0.341 ms, 6144: 0.492 -> 0.460).  Above: the MMD (quadratic in the cells) is the
        # longer of the two and keeps its head start beside the output GEMM; both kernels then end together (8192 cells:
        # 0.596 vs 0.605 ms; 16384: 1.38 vs 1.41).  The balance point moved when the likelihood kernel got faster
        # (DPP reductions, 146 -> 128 us alone): with the slower kernel the late fork won at 8192 cells (0.611 vs 0.632).
        # A first MMD row block beside the output GEMM and the rest forked behind it: 0.601-0.62 ms, dropped.
        # Replayed as a HIP graph the late fork wins up to 8192 cells (round 3, 300-step runs: 6144 cells 0.452 -> 0.426 ms,
        # 7168: 0.532 -> 0.495, 8192: 0.595 -> 0.568 - the graph executor submits the branch behind the likelihood kernel's
        # predecessor without the plan's marker in front of it, and the graph has no look-ahead gather to lose).
        after_max = MMD_AFTER_OUT_MAX_GRAPH if torch.cuda.is_current_stream_capturing() else MMD_AFTER_OUT_MAX
        after_out = (do_mmd and not mmd_split and compute_grad and self.overlap
                     and (MMD_AFTER_OUT == "1" or (MMD_AFTER_OUT == "auto" and B <= after_max)))
        if do_mmd:
            if mmd_split:
                nA = L.sca_mmd_rows_num_partials(B, n1, Z)
                self._n_part = self._grow_mmd_part(nA + L.sca_mmd_rows_num_partials(B, B - n1, Z))

                def mmd_work(sa):
                    self._mmd_launch(sa, ws["z"], Z, self._cond_cur, cond_p, B, compute_grad, rows=(0, n1))
            else:
                self._n_part = self._mmd_partials(B, Z)

                def mmd_work(sa):
                    self._mmd_launch(sa, ws["z"], Z, self._cond_cur, cond_p, B, compute_grad)
            # (the MMD's cap of three workgroups per CU keeps a slot free for kernels that arrive AFTER it; forked behind
            #  the likelihood kernel's predecessor it needs none: 0.632 ms with the cap, 0.612 without at 8192 cells)
            _lib.check(L.sca_mmd_reserve_lds(MMD_PAD_AFTER_OUT if after_out else MMD_PAD), "mmd_reserve_lds")
            if not after_out:
                self._defer(mmd_work)
        ev_side = None if after_out else self._flush()
        Hd = self.dec[-1].n_out
        h_last = ws[self.dec[-1].tag + ".h"].data_ptr()
        out_fused = "ext_ovf" in ws and "gam_tab" in ws and self.csr is not None and not mmd_split
        ev_mmd2 = None
        rpb = _recon_rows_per_block(B)
        gob = self.g[self.out_b].data_ptr() if T[self.out_b] else None
        if out_fused:
            # ONE launch: logits by MFMA into registers, softmax + likelihood + backward from there, only d-logits
            # written.  It needs the theta / gamma tables and the zeroed accumulators of the prologue.
            self._wait(self._ev_tables or self._ev_pro)
            if after_out:
                self._defer(mmd_work)
            ip, ix, dv = self.csr
            _lib.check(L.sca_out_nb_fused(h_last, Hd, P[self.out_w].data_ptr(), Hd, P[self.out_b].data_ptr(),
                                          ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), ptr(row_ids),
                                          ws["sf"].data_ptr(), cond_p, ws["tab_theta"].data_ptr(),
                                          ws["gam_tab"].data_ptr(), B, G, Hd, max(C, 1), ws["logits"].data_ptr(), G,
                                          ws["ext_ovf"].data_ptr(), G, ws["row_ll"].data_ptr(),
                                          ws["dthetaT"].data_ptr(), gob, int(compute_grad), st), "out_nb_fused")
            recon_sign = -1.0
        else:
            if after_out:
                self._wait(self._ev_tables or self._ev_pro)
            self._gemm(0, 0, h_last, Hd, P[self.out_w].data_ptr(), Hd, B, G, Hd, st, out=ws["logits"].data_ptr(), ldc=G,
                       bias=P[self.out_b].data_ptr(), force_s1=True)
            if after_out:
                self._defer(mmd_work)
            ev_mmd2 = None
            if mmd_split:
                ev_mmd2 = self._next_event()

                def mmd_work2(sa):
                    self._mmd_launch(sa, ws["z"], Z, self._cond_cur, cond_p, B, compute_grad, rows=(n1, B - n1), part_off=nA)
                    self._call(ev_mmd2.record, self._lane_stream(0))
            # (measured: with this wait moved in front of the output GEMM - to spare the stream operation between the GEMM
            #  and the likelihood kernel - the GEMM is dispatched behind the MMD, which then holds every CU: 0.665 -> 0.753 ms;
            #  in front of the MMD's fork event, so that the MMD is not ahead of the GEMM: +4 us, twice)
            if not after_out:
                self._wait(self._ev_tables or self._ev_pro)
            if self.recon == "mse":
                _lib.check(L.sca_mse_rows(ws["logits"].data_ptr(), G, ws["x"].data_ptr(), G, B, G, rpb,
                                          ws["row_ll"].data_ptr(), gob, int(compute_grad), 1, None, None, st), "mse_rows")
                recon_sign = 1.0
            else:
                ip, ix, dv = self.csr
                _lib.check(L.sca_recon_rows_tab(0, ws["logits"].data_ptr(), G, None, ip.data_ptr(), ix.data_ptr(),
                                                dv.data_ptr(), ptr(row_ids), ws["sf"].data_ptr(), cond_p,
                                                ws["tab_theta"].data_ptr(), ws["tab_lte"].data_ptr(), B, G, max(C, 1),
                                                rpb, ws["row_ll"].data_ptr(), ws["dthetaT"].data_ptr(), gob, None,
                                                int(compute_grad), None, None, ptr(ws.get("gam_tab")), st),
                           "recon_rows")
                recon_sign = -1.0
        if after_out:
            ev_side = self._flush()
        if mmd_split:
            self._defer(mmd_work2)
        self._defer_finalize(recon_sign, B, accumulate_logs)
        if not compute_grad:
            self._wait(self._flush())
            return self.losses
        # ---- backward
        dl = ws["logits"].data_ptr()

        def recon_side(sa):
            if T[self.out_w]:
                # (register-direct kernel; SCA_OUT_DW_KK=0 = the tiled GEMM, whose small workgroups leave the CUs to
                #  mid_bwd sooner: it won 2.5 us while the MMD forked before the output GEMM, and loses 6 us now)
                self._grad_gemm(1, 1, dl, G, h_last, Hd, G, Hd, B, sa, Gd[self.out_w],
                                allow_kk=os.environ.get("SCA_OUT_DW_KK", "1") == "1")
            if self.has_theta and T["theta"]:
                _lib.check(L.sca_transpose_scale(ws["dthetaT"].data_ptr(), max(C, 1), G, -1.0 / B,
                                                 Gd["theta"].data_ptr(), sa), "transpose_scale")
        # own stream: starts right behind the likelihood kernel (same dependency point as the loss finalisation)
        # (late fork: the MMD is still running when the likelihood kernel ends; with SCA_OUT_DW_AFTER_MMD=1 this
        #  weight-gradient GEMM - needed by the gradient tail only - waits for it, so that the dX GEMM, which the chain
        #  needs, shares the chip with the MMD alone)
        dw_after_mmd = after_out and do_mmd and not mmd_split and os.environ.get("SCA_OUT_DW_AFTER_MMD", "0") == "1"
        self._defer(recon_side, lane=1, reuse=not mmd_split and not dw_after_mmd, join=(0,) if dw_after_mmd else ())
        if mmd_split:
            self._flush()                       # the second MMD block must be enqueued before mid_bwd can wait for it
        sp, S, stride = self._gemm(0, 1, dl, G, P[self.out_w].data_ptr(), Hd, B, Hd, G, st, max_split=self._mid_max_split)
        if do_mmd:
            self._wait(ev_side)
            if ev_mmd2 is not None:
                self._call(torch.cuda.current_stream(self.device).wait_event, ev_mmd2)
        _lib.check(L.sca_mid_bwd(self._mid_args(B, cond_p, p_drop, eps, masks, sp, S, stride,
                                                ws["mmd_g1"].data_ptr() if do_mmd else None, pre0=pre0), st),
                   "mid_bwd")
        # weight-side gradients of every middle layer: one dependency point, aux stream
        mids = [(ly, ws["z"].data_ptr() if i == 0 else ws[self.dec[i - 1].tag + ".h"].data_ptr())
                for i, ly in enumerate(self.dec)]
        mids += [(ly, ws[self.enc[i].tag + ".h"].data_ptr()) for i, ly in enumerate(self.enc[1:])]
        Hl = self.H[-1]
        h_enc, dh = ws[self.enc[-1].tag + ".h"].data_ptr(), ws["dheads"].data_ptr()

        def mid_side(sa):
            for ly, a_prev in mids:
                da = ws[ly.tag + ".da"].data_ptr()
                if T[ly.w]:
                    self._grad_gemm(1, 1, da, ly.n_out, a_prev, ly.n_in, ly.n_out, ly.n_in, B, sa, Gd[ly.w])
                need_b, need_c = bool(ly.b) and T[ly.b], bool(ly.wc) and T[ly.wc]
                if need_b or need_c:
                    self._colsum(da, B, ly.n_out, cond_p, C, Gd[ly.wc].data_ptr() if need_c else None,
                                 Gd[ly.b].data_ptr() if need_b else None, sa)
            if any(T[k] for k in self.heads_names):
                self._grad_gemm(1, 1, dh, 2 * Z, h_enc, Hl, 2 * Z, Hl, B, sa, self.heads_gw)
                self._colsum(dh, B, 2 * Z, None, 0, None, self.heads_gb.data_ptr(), sa)
        # da of the first layer was written by mid_bwd: only its weight-side gradients remain
        sides = []
        self._layer_backward(self.enc[0], ws["x"].data_ptr(), None, 0, 0, None, False, B, cond_p, p_drop, masks, st,
                             have_da=True, side_out=sides)

        do_next = self._pf_flags[1]

        def weight_side(sa):      # the wide first-layer dY^T X: register-direct kernel on the aux stream
            self._gg_jobs = []
            try:
                for f in sides:
                    f(sa)
            finally:
                jobs, self._gg_jobs = self._gg_jobs, None
            self._launch_group(jobs, sa)        # (only when the first layer did not qualify for the kk kernel)
        def gather_next(sa):
            ip, ix, dv = self.csr
            _lib.check(L.sca_csr_gather_rows(ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), self._rows_next.data_ptr(),
                                             B, G, 0 if self.recon == "mse" else 1, ws["x"].data_ptr(), G,
                                             ws["sf"].data_ptr(), sa), "csr_gather[next]")
        self._defer(weight_side)
        # ... and beside it, on the main stream (idle after mid_bwd), ONE grouped launch for every narrow layer
        self._gg_jobs = []
        try:
            mid_side(st)
        finally:
            jobs, self._gg_jobs = self._gg_jobs, None
        self._launch_group(jobs, st)
        # (measured: gradient tail + Adam on the main stream instead, so that the next step queues behind Adam without
        #  a cross-stream hop: 0.6640 -> 0.6626 ms, within noise - the ~15 us between Adam and the next step's first
        #  kernel is dispatch latency, not the hop; the host enqueues a step in 0.39 ms and stays ahead.  Again with the
        #  first-layer weight gradient on the main stream too and the grouped GEMM on the aux stream: the two GEMMs then
        #  start together and take 64 / 61 us instead of 51 / 41, and the tail still starts 17 us behind them - the waits
        #  for the aux streams' events cost that whether or not the events have fired: 0.599 -> 0.604 ms)
        if not do_next:
            self._defer_tail(join=(1,))
        else:
            # The NEXT minibatch's gather (train_step(next_rows=...)): on the first aux stream, directly behind the last
            # reader of ws["x"] (the first-layer weight gradient above, same stream: no event needed), while the
            # gradient tail and Adam move to the second aux stream and run beside it - the step that follows starts at
            # its first GEMM.  Same kernel, same inputs as the in-line gather at the head of a step.
            # (An event recorded on the first aux stream behind that GEMM and waited for on the second - gather there,
            #  tail staying here - crashed hipStreamEndCapture when the step was captured into a HIP graph.)
            self._defer_tail(join=(0,), lane=1)
            self._defer(gather_next, lane=0, reuse=True)
        self._wait(self._flush())
        return self.losses

    # ------------------------------------------------------------------------------------------ scPoli
    def _embed_lookup(self, covs, B, st):
        """cvec [B, E] = hstack_i embeddings[i][covs[:, i]] with in-place max_norm renormalisation of touched rows."""
        L, ws = self.lib, self.ws
        if not self.emb:
            return
        assert covs is not None and covs.dtype == torch.int32 and covs.is_contiguous() and covs.shape[1] == len(self.emb)
        col = 0
        for i, (k, n, e) in enumerate(self.emb):
            _lib.check(L.sca_embed_lookup(self.p[k].data_ptr(), n, e, covs.data_ptr() + 4 * i, len(self.emb), B,
                                          self.max_norm, ws["emb_flags"].data_ptr(), ws["cvec"].data_ptr() + 4 * col,
                                          self.E, st), "embed_lookup")
            col += e

    def _embed_scatter(self, B, sa):
        """Embedding gradients: scatter-add of dcvec (encoder + decoder injection sites) on the aux stream."""
        L, ws, Gd, T = self.lib, self.ws, self.g, self.trainable
        covs = self._covs_cur
        ga = ws["dcvec_enc"].data_ptr() if self.enc[0].wc else None
        gb = ws["dcvec_dec"].data_ptr() if self.dec[0].wc else None
        col = 0
        for i, (k, n, e) in enumerate(self.emb):
            if T[k]:
                self._zero(Gd[k], sa)
                _lib.check(L.sca_embed_scatter_add(None if ga is None else ga + 4 * col,
                                                   None if gb is None else gb + 4 * col, self.E, B, e, n,
                                                   covs.data_ptr() + 4 * i, len(self.emb), Gd[k].data_ptr(), sa),
                           "embed_scatter")
            col += e

    def _launch_proto(self, proto, B, compute_grad):
        """eta * (labeled + unlabeled_weight * unlabeled) prototype loss on z (trainers/scpoli/trainer.py:542-575) on the
        aux stream; its value lands in the third loss slot, its gradient joins dz in latent_bwd."""
        L, ws = self.lib, self.ws
        Pl, Pu = proto.get("P"), proto.get("Pu")
        if Pl is None and Pu is None:
            return 0
        K = 0 if Pl is None else Pl.shape[0]
        Ku = 0 if Pu is None else Pu.shape[0]
        need = L.sca_proto_workspace_bytes(B, K, Ku)
        if "proto_work" not in ws or ws["proto_work"].numel() < need:
            if "proto_work" in ws:
                self._invalidate()
            ws["proto_work"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        ct, lab = proto.get("celltypes"), proto.get("labeled")
        eta, uw = float(proto.get("eta", 1.0)), float(proto.get("unlabeled_weight", 0.0))

        def work(sa):
            _lib.check(L.sca_proto_loss(ws["z"].data_ptr(), self.Zt, B, self.Zt, _lib.ptr(Pl), K, _lib.ptr(ct),
                                        0 if ct is None else ct.shape[1], _lib.ptr(lab), eta, _lib.ptr(Pu), Ku,
                                        eta * uw, ws["proto_work"].data_ptr(), ws["mmd_part"].data_ptr(),
                                        ws["proto_raw"].data_ptr(), 1,
                                        ws["proto_dz"].data_ptr() if compute_grad else None, self.Zt, None, sa),
                       "proto")
        self._defer(work)
        if compute_grad:
            self._dz2_cur = (ws["proto_dz"].data_ptr(), 1)
        return 1

    # ------------------------------------------------------------------------------------------ expiMap
    def _expimap_pack(self, st):
        """Weff [G][Zt] = [expr_L.weight*mask | ext_L_m.weight*ext_mask | ext_L.weight] (column order of
        MaskedCondLayers.forward)."""
        L, P, ws = self.lib, self.p, self.ws
        Zt, col = self.Zt, 0
        for (k, w, _), md in zip(self.xp_parts, self.xp_masks):
            _lib.check(L.sca_masked_copy(P[k].data_ptr(), w, _lib.ptr(md), w, ws["Weff"].data_ptr() + 4 * col, Zt,
                                         self.G, w, st), "masked_copy")
            col += w

    def _expimap_tail(self, row_ids, cond_p, B, compute_grad, accumulate_logs, st):
        """Decoder + losses + backward of expiMap after the latent block (models/expimap/expimap.py:180-222)."""
        L, P, Gd, ws, T = self.lib, self.p, self.g, self.ws, self.trainable
        G, C, Zt = self.G, self.C, self.Zt
        ptr = _lib.ptr
        AUX = "slab_aux" if self.overlap else "slab"
        # HSIC between annotated and unconstrained latent columns, beside the decoder (aux stream)
        if self.use_hsic:
            def hsic_work(sa):
                if compute_grad:
                    self._zero(ws["hsic_dz"], sa)
                side = ws["hsic_side"]
                for i, (sig_a, sig_b) in enumerate(self._hsic_sigmas):
                    _lib.check(L.sca_hsic_fwd_bwd(ws["z"].data_ptr(), Zt, B, Zt, side[i].data_ptr(), sig_a, sig_b,
                                                  self.hsic_beta, ws["hsic_work"].data_ptr(),
                                                  ws["mmd_part"].data_ptr(), ws["hsic_raw"].data_ptr(), int(i > 0),
                                                  ws["hsic_dz"].data_ptr() if compute_grad else None, Zt, sa), "hsic")
            self._defer(hsic_work)
            self._n_part = 1
        self._expimap_pack(st)
        coff = ws["coffT"].data_ptr() if self.xp_wc else None
        mk = self.masked
        if mk is not None:
            # tile-skipping masked decoder: compact weights of this step, then K_t (not Zt) k-steps per gene tile
            mk_meta = (mk["tile_k"].data_ptr(), mk["tile_cols"].data_ptr(), mk["NT"], mk["Kmax"])
            _lib.check(L.sca_masked_pack(ws["Weff"].data_ptr(), Zt, *mk_meta, G, Zt, ws["mk_Wc"].data_ptr(), st),
                       "masked_pack")
            _lib.check(L.sca_masked_fwd(ws["z"].data_ptr(), Zt, ws["mk_Wc"].data_ptr(), *mk_meta, B, G, Zt,
                                        ws["logits"].data_ptr(), G, st), "masked_fwd")
        else:
            self._gemm(0, 0, ws["z"].data_ptr(), Zt, ws["Weff"].data_ptr(), Zt, B, G, Zt, st,
                       out=ws["logits"].data_ptr(), ldc=G, force_s1=True)
        ev_side = self._flush()
        lenc = self.lenc if self.lenc_active else None
        sf_p = ws["sf"].data_ptr()
        if lenc is not None:        # size factor = exp(sampled library latent) (expimap.py:186-187)
            sf_p = self._lenc_forward(B, cond_p, st)
        self._wait(self._ev_tables or self._ev_pro)
        rpb = _recon_rows_per_block(B)
        if self.recon == "mse":
            _lib.check(L.sca_mse_rows(ws["logits"].data_ptr(), G, ws["x"].data_ptr(), G, B, G, rpb,
                                      ws["row_ll"].data_ptr(), None, int(compute_grad), self.last_act, cond_p, coff,
                                      st), "mse_rows")
            recon_sign = 1.0
        elif self.last_layer != "softmax":
            ip, ix, dv = self.csr
            _lib.check(L.sca_nb_elem_rows(self.last_act, ws["logits"].data_ptr(), G, ip.data_ptr(), ix.data_ptr(),
                                          dv.data_ptr(), ptr(row_ids), cond_p, ws["tab_theta"].data_ptr(),
                                          ws["tab_lte"].data_ptr(), B, G, max(C, 1), rpb, ws["row_ll"].data_ptr(),
                                          ws["dthetaT"].data_ptr(), int(compute_grad), coff, st), "nb_elem_rows")
            recon_sign = -1.0
        else:
            ip, ix, dv = self.csr
            _lib.check(L.sca_recon_rows_tab(0, ws["logits"].data_ptr(), G, None, ip.data_ptr(), ix.data_ptr(),
                                            dv.data_ptr(), ptr(row_ids), sf_p, cond_p,
                                            ws["tab_theta"].data_ptr(), ws["tab_lte"].data_ptr(), B, G, max(C, 1), rpb,
                                            ws["row_ll"].data_ptr(), ws["dthetaT"].data_ptr(), None, None,
                                            int(compute_grad), coff,
                                            ws["dsf"].data_ptr() if (lenc is not None and compute_grad) else None,
                                            ptr(ws.get("gam_tab")), st),
                       "recon_rows")
            recon_sign = -1.0
        self._defer_finalize(recon_sign, B, accumulate_logs)
        if not compute_grad:
            self._wait(self._flush())
            return self.losses
        # ---- backward: weight-side gradients on the aux stream, dz on the main stream
        dl = ws["logits"].data_ptr()

        def side(sa):
            if self.has_theta and T["theta"]:
                _lib.check(L.sca_transpose_scale(ws["dthetaT"].data_ptr(), max(C, 1), G, -1.0 / B,
                                                 Gd["theta"].data_ptr(), sa), "transpose_scale")
            if any(T[k] for (k, _, _) in self.xp_parts):
                if mk is not None:
                    S_w = L.sca_masked_dw_slabs(B, mk["NT"])
                    _lib.check(L.sca_masked_dw(ws["z"].data_ptr(), Zt, dl, G, *mk_meta, B, G, Zt,
                                               ws["mk_dWc"].data_ptr(), mk["n"], sa), "masked_dw")
                    self._zero(ws["dWeff"], sa)
                    _lib.check(L.sca_masked_unpack(ws["mk_dWc"].data_ptr(), S_w, mk["n"], *mk_meta, G, Zt,
                                                   ws["dWeff"].data_ptr(), Zt, sa), "masked_unpack_tiles")
                else:
                    self._gemm_to(1, 1, dl, G, ws["z"].data_ptr(), Zt, G, Zt, B, sa, ws["dWeff"], slab_key=AUX)
                col = 0
                for (k, w, _), md in zip(self.xp_parts, self.xp_masks):
                    if T[k]:
                        _lib.check(L.sca_masked_copy(ws["dWeff"].data_ptr() + 4 * col, Zt, ptr(md), w,
                                                     Gd[k].data_ptr(), w, G, w, sa), "masked_unpack")
                    col += w
            if self.xp_wc and T[self.xp_wc]:
                self._colsum(dl, B, G, cond_p, C, Gd[self.xp_wc].data_ptr(), None, sa)
        self._defer(side)
        if lenc is not None:
            self._lenc_backward(B, cond_p, st)
        if mk is not None:
            sp, S, stride = ws["mk_dz"].data_ptr(), L.sca_masked_dx_slabs(B, mk["NT"]), B * Zt
            _lib.check(L.sca_masked_dx(dl, G, ws["mk_Wc"].data_ptr(), *mk_meta, B, G, Zt, sp, stride, st), "masked_dx")
        else:
            sp, S, stride = self._gemm(0, 1, dl, G, ws["Weff"].data_ptr(), Zt, B, Zt, G, st)
        if self.use_hsic:
            self._wait(ev_side)
        self._backward_latent_encoder(B, cond_p, self._p_drop_cur, self._masks_cur, st, sp, S, stride,
                                      ws["hsic_dz"].data_ptr() if self.use_hsic else None,
                                      1 if self.use_hsic else 0)
        return self.losses

    def _lenc_forward(self, B, cond_p, st):
        """expiMap library-size encoder (models/expimap/expimap.py:164-171): ExtEncoder([G, 128], 1) on the
        log-transformed minibatch, sampled like z, size factor = exp(sample) (expimap.py:186-187).  Returns the
        device pointer of the [B] size factors."""
        L, P, ws, ly = self.lib, self.p, self.ws, self.lenc
        masks, p_drop = self._masks_cur, self._p_drop_cur
        sp, S, stride = self._gemm(0, 0, ws["x"].data_ptr(), ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out, ly.n_in,
                                   st)
        h = self._layer_block_fwd(ly, sp, S, stride, B, (self.C, cond_p, None, 0), p_drop,
                                  _lib.ptr(masks.get(ly.mask_key)), True, self.step_ctr.data_ptr(), st)
        Hl = ly.n_out
        self._gemm(0, 0, h, Hl, self.lheads_w.data_ptr(), Hl, B, 2, Hl, st, out=ws["l_mu_lv"].data_ptr(), ldc=2,
                   bias=self.lheads_b.data_ptr(), force_s1=True)
        _lib.check(L.sca_latent_fwd(ws["l_mu_lv"].data_ptr(), 1, 0, None, B, 1, _lib.ptr(masks.get("eps_l")), self.seed,
                                    3, self.step_ctr.data_ptr(), 1, ws["l_mu_lv"].data_ptr(), ws["l_eps"].data_ptr(),
                                    ws["l_z"].data_ptr(), ws["l_kl"].data_ptr(), st), "latent_fwd[lenc]")
        _lib.check(L.sca_exp_f32(ws["l_z"].data_ptr(), ws["sf_l"].data_ptr(), B, st), "exp[lenc]")
        return ws["sf_l"].data_ptr()

    def _lenc_backward(self, B, cond_p, st):
        """Gradient of the reconstruction loss through the sampled size factor into the library-size encoder: no KL
        term is attached to that latent (expimap.py:198-202 only penalises z1)."""
        L, P, Gd, ws, T, ly = self.lib, self.p, self.g, self.ws, self.trainable, self.lenc
        Hl = ly.n_out
        _lib.check(L.sca_mul_f32(ws["dsf"].data_ptr(), ws["sf_l"].data_ptr(), ws["l_dz"].data_ptr(), B, st), "mul[lenc]")
        _lib.check(L.sca_latent_bwd(ws["l_dz"].data_ptr(), 1, B, None, 0, 0, ws["l_mu_lv"].data_ptr(),
                                    ws["l_eps"].data_ptr(), self.zero_hyper.data_ptr(), B, 1,
                                    ws["l_dheads"].data_ptr(), st), "latent_bwd[lenc]")
        dh, h = ws["l_dheads"].data_ptr(), ws["lenc0.h"].data_ptr()
        if any(T[k] for k in self.lheads_names):
            def heads_side(sa):
                self._grad_gemm(1, 1, dh, 2, h, Hl, 2, Hl, B, sa, self.lheads_gw)
                self._colsum(dh, B, 2, None, 0, None, self.lheads_gb.data_ptr(), sa)
            self._defer(heads_side)
        sp, S, stride = self._gemm(0, 1, dh, 2, self.lheads_w.data_ptr(), Hl, B, Hl, 2, st)
        self._layer_backward(ly, ws["x"].data_ptr(), sp, S, stride, None, False, B, cond_p, self._p_drop_cur,
                             self._masks_cur, st)

    def _scratch_bh(self, B, Hd):
        if not hasattr(self, "_sbh") or self._sbh.numel() < B * Hd:
            self._invalidate()
            self._sbh = torch.empty(B * Hd, dtype=torch.float32, device=self.device)
        return self._sbh[:B * Hd].view(B, Hd)

    # ------------------------------------------------------------------------------------------ optimiser
    def trainable_ranges(self, weight_decay=0.0, wd_overrides: Optional[Dict[str, float]] = None):
        """Contiguous [start, end, wd) ranges of the flat buffer covering the trainable tensors: tensors that follow
        each other in the buffer (only alignment padding between them) and share a weight decay are merged; a frozen
        tensor always splits."""
        wd_overrides = wd_overrides or {}
        items = sorted((self.offsets[k], self.offsets[k] + int(np.prod(self.shapes[k])), self.trainable[k],
                        float(wd_overrides.get(k, weight_decay))) for k in self.names)
        merged, open_wd = [], None
        for s, e, t, wd in items:
            if not t:
                open_wd = None
                continue
            if open_wd is not None and open_wd == wd:
                merged[-1][1] = e
            else:
                merged.append([s, e, wd])
                open_wd = wd
        return [(s, e, wd) for s, e, wd in merged]

    def adam_step(self, weight_decay=0.04, wd_overrides: Optional[Dict[str, float]] = None, stream=None):
        """One torch.optim.Adam-equivalent update of every trainable tensor; hyper[] must hold lr / bias corrections
        for step t = self.adam_t + 1 (call begin_step first).  wd_overrides: per-tensor weight decay (scPoli keeps its
        embeddings in a weight_decay=0 parameter group, trainers/scpoli/trainer.py:297-305)."""
        L = self.lib
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        for s, e, wd in self.trainable_ranges(weight_decay, wd_overrides):
            _lib.check(L.sca_adam_step(self.flat_p[s:].data_ptr(), self.flat_g[s:].data_ptr(),
                                       self.flat_m[s:].data_ptr(), self.flat_v[s:].data_ptr(), e - s,
                                       self.hyper.data_ptr(), 0.9, 0.999, wd, st), "adam")

    # ------------------------------------------------------------------------------------------ data-parallel MMD
    def set_dp(self, rank: int, world: int, exact_mmd: bool = True, gather=None):
        """Data-parallel context of this engine.  With `exact_mmd` the MMD of a step is the reference's MMD over the
        GLOBAL minibatch (scarches/models/trvae/losses.py:231-270 couples every pair of cells): the ranks all-gather
        their latent rows + conditions (B_global x Z floats), every rank evaluates the full kernel sum - the value is
        then identical on all ranks - and keeps the gradient rows of its own cells, pre-scaled by n_global / n_local
        so that the weighted gradient all-reduce (weight n_local / n_global) adds the MMD term with weight 1.
        Cost: O(B_global^2) pair evaluations PER RANK; meant for the reference's batch sizes (<= a few thousand
        cells per global minibatch).  Without it each rank takes the MMD of its own slice (a per-rank estimator).
        `gather(out, local)` replaces torch.distributed.all_gather_into_tensor (tests)."""
        self._dp = dict(rank=int(rank), world=int(world), exact=bool(exact_mmd) and world > 1, gather=gather)
        self._invalidate()

    def _dp_exact(self):
        return self._dp is not None and self._dp["exact"] and self.use_mmd and self.C > 0

    def _dp_bufs(self, n_max, Dm):
        key = (n_max, Dm)
        b = self._dp_ws.get(key)
        if b is None:
            dev, Bg = self.device, self._dp["world"] * n_max
            ns = self.lib.sca_mmd_rows_num_slabs(Bg, n_max, Dm)
            b = dict(zpad=torch.zeros(n_max, Dm, device=dev),
                     cpad=torch.full((n_max,), -1, dtype=torch.int32, device=dev),
                     zall=torch.empty(Bg, Dm, device=dev), call=torch.empty(Bg, dtype=torch.int32, device=dev),
                     slabs=torch.empty(max(1, ns) * n_max * Dm, device=dev))   # grown for ragged slices if needed
            self._dp_ws[key] = b
        return b

    def _dp_gather(self, out, local):
        if self._dp_dry:
            # warm-up / plan-recording pass: NO collective (ranks record their plans at different steps - ragged last
            # minibatches give them different local sizes - so a collective here would pair with nothing and hang the
            # job); the rank's own rows stand in for the gathered minibatch, the pass's results are discarded
            n = local.shape[0]
            if out.dtype.is_floating_point:
                out.zero_()
            else:
                out.fill_(-1)          # condition -1 = padding row, ignored by the kernels
            r = self._dp["rank"]
            out[r * n:(r + 1) * n].copy_(local)
            return
        if self._dp["gather"] is not None:
            self._dp["gather"](out, local)
        else:
            from . import distributed as sdist
            sdist.all_gather_rows_(out, local)

    def _dp_stage_gather(self, b, feat, cond_t, B, Dm, stream):
        """Pad this rank's rows to n_max (condition -1 = ignored by the kernels) and all-gather rows + conditions."""
        with torch.cuda.stream(stream):
            zp, cp = b["zpad"], b["cpad"]
            if B < zp.shape[0]:
                zp[B:].zero_()
                cp[B:].fill_(-1)
            if B > 0:
                zp[:B].copy_(feat.view(-1)[:B * Dm].view(B, Dm))
                cp[:B].copy_(cond_t.view(-1)[:B])
            self._dp_gather(b["zall"], zp)
            self._dp_gather(b["call"], cp)

    def dp_idle_step(self, n_global):
        """A rank whose slice of a (tiny, last) minibatch is empty still takes part in the step's all-gathers."""
        if not self._dp_exact():
            return
        Dm = self.Zt if (self.mmd_on == "z" or self.arch == "expimap") else self.dec[0].n_out
        b = self._dp_bufs(-(-int(n_global) // self._dp["world"]), Dm)
        self._dp_stage_gather(b, None, None, 0, Dm, torch.cuda.current_stream(self.device))

    def _mmd_launch(self, sa, feat, Dm, cond_t, cond_p, B, compute_grad, rows=None, part_off=0):
        """MMD forward (+ gradient w.r.t. the feature rows into ws['mmd_g1']) on stream `sa`.  rows = (lo, n): only
        that row block of the minibatch (prepared weight table required; loss partials from part_off on)."""
        L, ws, C = self.lib, self.ws, self.C
        bnd = -1 if self.boundary is None else int(self.boundary)
        if rows is not None:
            lo, n = rows
            ns = L.sca_mmd_rows_num_slabs(B, n, Dm)
            if ns * n * Dm > ws["mmd_g"].numel():
                raise RuntimeError("MMD gradient slabs exceed the workspace (%d x %d x %d)" % (ns, n, Dm))
            dst = ws["mmd_g1"].data_ptr() + 4 * lo * Dm
            g_out = dst if ns == 1 else ws["mmd_g"].data_ptr()
            _lib.check(L.sca_mmd_fwd_bwd_prepared(feat.data_ptr(), Dm, cond_p, B, Dm, C, ws["Wt"].data_ptr(), lo, n,
                                                  ws["mmd_part"].data_ptr() + 8 * part_off,
                                                  g_out if compute_grad else None, n * Dm, sa), "mmd[rows]")
            if compute_grad and ns > 1:
                _lib.check(L.sca_splitk_reduce(ws["mmd_g"].data_ptr(), ns, n * Dm, dst, n * Dm, sa), "mmd_reduce")
            return
        if not self._dp_exact():
            ns = L.sca_mmd_num_slabs(B, Dm)
            if ns * B * Dm > ws["mmd_g"].numel():
                raise RuntimeError("MMD gradient slabs exceed the workspace (%d x %d x %d)" % (ns, B, Dm))
            g_out = ws["mmd_g1"] if ns == 1 else ws["mmd_g"]          # a single slab is the gradient itself
            if getattr(self, "_mmd_prepared", False):                  # weight table built in the prologue
                _lib.check(L.sca_mmd_fwd_bwd_prepared(feat.data_ptr(), Dm, cond_p, B, Dm, C, ws["Wt"].data_ptr(), 0, B,
                                                      ws["mmd_part"].data_ptr(),
                                                      g_out.data_ptr() if compute_grad else None, B * Dm, sa),
                           "mmd[prepared]")
            else:
                _lib.check(L.sca_mmd_fwd_bwd(feat.data_ptr(), Dm, cond_p, B, Dm, C, bnd, self.beta,
                                             ws["Wt"].data_ptr(), ws["counts"].data_ptr(), ws["mmd_part"].data_ptr(),
                                             g_out.data_ptr() if compute_grad else None, B * Dm, sa), "mmd")
            if compute_grad and ns > 1:                                # gradient slabs summed off the critical path
                _lib.check(L.sca_splitk_reduce(ws["mmd_g"].data_ptr(), ns, B * Dm,
                                               ws["mmd_g1"].data_ptr(), B * Dm, sa), "mmd_reduce")
            return
        d = self._dp
        n_global = int(self._dp_n_global or B * d["world"])
        n_max = -(-n_global // d["world"])
        Bg = d["world"] * n_max
        b = self._dp_bufs(n_max, Dm)
        stream = self.aux if (self.overlap and sa == self.aux.cuda_stream) else torch.cuda.current_stream(self.device)
        self._call(self._dp_stage_gather, b, feat, cond_t, B, Dm, stream)
        # this rank's row block against all columns of the gathered minibatch.  beta carries n_global / n_local: the
        # logged MMD is then (n_global / n_local) x the block's share and the gradient rows are scaled alike, so the
        # n_local / n_global-weighted all-reduces (gradients per step, log sums per epoch) give the whole-minibatch
        # MMD and its gradient with weight 1.
        ns = L.sca_mmd_rows_num_slabs(Bg, B, Dm)
        if compute_grad and ns > 1 and b["slabs"].numel() < ns * B * Dm:      # eager step: allocation is fine here
            b["slabs"] = torch.empty(ns * B * Dm, device=self.device)
        g_out = ws["mmd_g1"].data_ptr() if ns == 1 else b["slabs"].data_ptr()
        _lib.check(L.sca_mmd_rows_fwd_bwd(b["zall"].data_ptr(), Dm, b["call"].data_ptr(), Bg, Dm, C, bnd,
                                          self.beta * (n_global / float(B)), ws["Wt"].data_ptr(),
                                          ws["counts"].data_ptr(), d["rank"] * n_max, B, ws["mmd_part"].data_ptr(),
                                          g_out if compute_grad else None, B * Dm, sa), "mmd_rows[dp]")
        if compute_grad and ns > 1:
            _lib.check(L.sca_splitk_reduce(b["slabs"].data_ptr(), ns, B * Dm, ws["mmd_g1"].data_ptr(), B * Dm, sa),
                       "mmd_reduce[dp]")

    def _grow_mmd_part(self, n):
        if self.ws["mmd_part"].numel() < n:
            self._invalidate()
            self.ws["mmd_part"] = torch.zeros(n, dtype=torch.float64, device=self.device)
        return n

    def _mmd_partials(self, B, Dm):
        """Number of loss partials the MMD launch of this step writes (ws['mmd_part'] is grown to hold them)."""
        if self._dp_exact():
            d = self._dp
            Bg = d["world"] * -(-int(self._dp_n_global or B * d["world"]) // d["world"])
            n = self.lib.sca_mmd_rows_num_partials(Bg, B, Dm)
        else:
            n = self.lib.sca_mmd_num_partials(B, Dm)
        if self.ws["mmd_part"].numel() < n:
            self._invalidate()
            self.ws["mmd_part"] = torch.zeros(n, dtype=torch.float64, device=self.device)
        return n

    # ------------------------------------------------------------------------------------------ whole step
    def train_step(self, row_ids, cond, B, alpha_kl, lr, eps=0.01, clip=0.0, weight_decay=0.04, world=1,
                   use_graph="auto", wd_overrides=None, grad_weight=None, n_global=None, next_rows=None):
        """One optimisation step: forward + backward (+ gradient all-reduce) + Adam.

        All per-step scalars live in device memory (`hyper`, `step_ctr`) and the minibatch ids are copied into static
        buffers, so the launch sequence of a given batch size never changes.  It is recorded once and then either
          * "plan" : re-issued from the recorded (function, arguments) list - no Python work per launch besides the
                     ctypes call itself, launches stay in the hand-ordered main-first sequence;
                     "auto" for B > 8192 and for every data-parallel step (10240 cells: 0.757 ms against 0.784 as a
                     graph; the look-ahead loader below exists in this mode only);
          * True   : replayed as HIP graphs (one graph launch per step; best when launch-bound: 0.27 ms at B=128
                     against 0.37 ms eager) - "auto" for B <= 8192 on one GPU.  Round 3 (300-step runs, plan / graph):
                     2048 cells 0.255 / 0.226 ms, 4096: 0.332 / 0.310, 6144: 0.438 / 0.426, 8192: 0.573 / 0.568 with
                     the MMD forked behind the output GEMM inside graphs (_fused_path); the plan is host-bound up to
                     ~5000 cells (0.25 - 0.33 ms of host time per step against 0.13 - 0.24 for a graph launch);
          * False  : launched eagerly through the normal Python path.

        `next_rows` (int64 device tensor, same size as `row_ids`): the row ids of the minibatch the NEXT call will
        train on.  The recorded step then ends with that minibatch's CSR gather - on an aux stream, beside the gradient
        tail and Adam - and the next call, given the same tensor (same storage, unmodified), starts at its first GEMM
        instead of the gather: the loader runs ahead of the model as the reference's DataLoader workers do.  Results
        are those of the un-pipelined sequence (same kernel, same inputs, same order of the arithmetic)."""
        if use_graph == "auto":
            # (world > 1: the gradient all-reduce is a host call between the two replayed halves of the step.  The
            # intermittent hangs of bench.py's two-rank one-GPU rehearsal came from ProcessGroupGloo's device-tensor
            # path, not from the launch mode: see distributed._host_staged.)
            use_graph = True if (B <= GRAPH_MAX_B and world <= 1) else "plan"
        self._dp_world, self._dp_grad_weight = int(world), grad_weight
        if world > 1 and use_graph is True:
            use_graph = "plan"      # the all-reduce is a host call inside the recorded sequence (not graph-captured)
        dp_exact = world > 1 and self._dp_exact()
        if dp_exact:
            self._dp_n_global = int(n_global) if n_global is not None else B * world
            # The step holds two all-gathers in its middle: they are host calls inside the recorded launch plan (not
            # graph-capturable).  Every rank issues exactly two gathers + one all-reduce per step whatever the state
            # of its plans: warm-up and recording passes gather nothing (_dp_dry), so ranks may record at different
            # steps (ragged last minibatches give them different local sizes).
            if use_graph is True:
                use_graph = "plan"
        if use_graph is False:
            self.begin_step(alpha_kl, lr, eps, clip)
            self._pf_flags = (False, False)
            self._step_and_adam(row_ids, cond, B, weight_decay, wd_overrides, world > 1)
            return self.losses
        def plan_key():
            return (B, use_graph is True, float(weight_decay), tuple(sorted((wd_overrides or {}).items())), world > 1,
                    tuple(sorted(k for k, v in self.trainable.items() if v)), self.use_mmd, self.mmd_on,
                    self.boundary, self.beta, torch.cuda.current_stream(self.device).cuda_stream,
                    self._dp_n_global if dp_exact else None, self.hsic_beta, self.dr_rate, self.seed, self.freeze,
                    self._ws_gen, self._pf_flags)
        self._ensure(B)
        # (launch plans only: captured into a HIP graph the trailing gather costs more than the head gather it
        #  replaces - 0.350 -> 0.360 ms at 4096 cells, 0.239 -> 0.245 ms at 1024)
        do_next = (next_rows is not None and (use_graph == "plan" or (use_graph is True and GRAPH_PREFETCH))
                   and self._pf_possible(next_rows, B))
        if do_next and (self._rows_next is None or self._rows_next.numel() < B):
            torch.cuda.synchronize(self.device)
            self._rows_next = torch.zeros(max(B, self.Bcap), dtype=torch.int64, device=self.device)
            self._invalidate()
        have_cur = self._pf_matches(row_ids, B)
        self._pf_flags = (have_cur, do_next)
        try:
            return self._train_step_planned(row_ids, cond, B, alpha_kl, lr, eps, clip, weight_decay, world, use_graph,
                                            wd_overrides, grad_weight, next_rows, plan_key, have_cur, do_next)
        finally:
            self._pf_flags = (False, False)

    def _train_step_planned(self, row_ids, cond, B, alpha_kl, lr, eps, clip, weight_decay, world, use_graph,
                            wd_overrides, grad_weight, next_rows, plan_key, have_cur, do_next):
        pf_stage = (next_rows, self._rows_next, B) if do_next else None
        key = plan_key()
        plans = self._plans
        g = plans.get(key)
        fast = (g is not None and row_ids.dtype == torch.int64 and cond.dtype == torch.int32 and row_ids.is_cuda
                and cond.is_cuda and row_ids.is_contiguous() and cond.is_contiguous()
                and row_ids.numel() == B and cond.numel() == B)
        if fast:        # scalars + minibatch staging in one launch
            self.begin_step(alpha_kl, lr, eps, clip, stage=(row_ids, g["rows"], cond, g["cond"], B), stage_next=pf_stage)
        else:
            self.begin_step(alpha_kl, lr, eps, clip)
            if g is None:
                g = (self._capture if use_graph is True else self._record)(B, weight_decay, wd_overrides, world > 1)
                # buffers allocated lazily during the warm-up pass bump the workspace generation (and drop the old
                # plans): file the new plan under the generation it was recorded against
                self._pf_flags = (have_cur, do_next)
                self._plans[plan_key()] = g
            g["rows"].copy_(row_ids, non_blocking=True)
            g["cond"].copy_(cond, non_blocking=True)
            if do_next:
                self._rows_next[:B].copy_(next_rows, non_blocking=True)
            if have_cur:
                # recording executes the step - trailing gather included - with placeholder rows: put this
                # minibatch back into ws["x"] (the replay below does not gather it)
                ip, ix, dv = self.csr
                _lib.check(self.lib._lib.sca_csr_gather_rows(
                    ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), g["rows"].data_ptr(), B, self.G,
                    0 if self.recon == "mse" else 1, self.ws["x"].data_ptr(), self.G, self.ws["sf"].data_ptr(),
                    torch.cuda.current_stream(self.device).cuda_stream), "csr_gather")
        if use_graph is True:
            g["fwd"].replay()
        else:
            self._replay(g["fwd"])
        # what ws["x"] holds once this step's launches have run: the next minibatch, or nothing to rely on
        self._pf = (dict(rows=next_rows, ptr=next_rows.data_ptr(), version=next_rows._version, B=B, gen=self._ws_gen,
                         csr=self.csr[0].data_ptr()) if do_next else None)
        return self.losses

    def _pf_possible(self, next_rows, B):
        """The trailing gather exists in the fused trVAE step with overlap on (its place in the stream schedule is
        _fused_path's): everything else keeps the in-line gather at the head of the step."""
        return (self.overlap and self.use_fused_mid and not (self.use_mmd and self.mmd_on != "z") and self.csr is not None
                and self.arch != "expimap" and not self.cond_embed
                and next_rows.is_cuda and next_rows.dtype == torch.int64 and next_rows.is_contiguous()
                and next_rows.numel() == B and os.environ.get("SCA_PREFETCH", "1") == "1")

    def _pf_matches(self, row_ids, B):
        """True when ws["x"] / ws["sf"] already hold this minibatch: the previous train_step gathered `next_rows` and
        `row_ids` is that tensor's storage, unmodified since (the engine holds a reference to it, so the address cannot
        have been re-used by another tensor; in-place writes bump the shared version counter)."""
        pf = self._pf
        return (pf is not None and pf["B"] == B and pf["gen"] == self._ws_gen and self.csr is not None
                and pf["csr"] == self.csr[0].data_ptr() and row_ids.dtype == torch.int64
                and row_ids.data_ptr() == pf["ptr"] and row_ids.numel() == B and row_ids.is_contiguous()
                and row_ids._version == pf["version"] and pf["rows"]._version == pf["version"])

    def _step_and_adam(self, rows, cond, B, weight_decay, wd_overrides, split):
        """forward + backward, the gradient all-reduce of a data-parallel step (`split`) and Adam, the last two queued
        right behind the gradient tail on its aux stream."""
        self._adam_after_tail = (weight_decay, wd_overrides)
        self._dp_split = bool(split)
        self._adam_done = False
        try:
            self.forward_backward(rows, cond, B)
            if not self._adam_done:                 # paths without a deferred tail (overlap off)
                if split:
                    self._call(self._dp_allreduce_main)
                self.adam_step(weight_decay, wd_overrides)
        finally:
            self._adam_after_tail = None
            self._dp_split = False

    def _dp_allreduce_main(self):
        if self._dp_dry or self._dp_world <= 1:
            return
        from . import distributed as sdist
        sdist.allreduce_mean_(self.flat_g, self._dp_world, weight=self._dp_grad_weight)

    @staticmethod
    def _replay(plan):
        for fn, args, checked in plan:
            rc = fn(*args)
            if checked and rc:
                _lib.check(rc, "replayed launch")

    def _warm(self, B, weight_decay, wd_overrides):
        """Static minibatch buffers + one throw-away step (workspaces, lazy module loads) that leaves parameters,
        Adam state and log accumulators untouched."""
        self._ensure(B)
        rows = torch.zeros(B, dtype=torch.int64, device=self.device)
        cond = torch.zeros(B, dtype=torch.int32, device=self.device)
        snap = (self.flat_p.clone(), self.flat_m.clone(), self.flat_v.clone(), self.accum.clone(),
                {k: v.clone() for k, v in self.buffers.items()})
        self._dp_dry = True
        try:
            self._step_and_adam(rows, cond, B, weight_decay, wd_overrides, False)
        finally:
            self._dp_dry = False
        torch.cuda.synchronize(self.device)
        return rows, cond, snap

    def _restore(self, snap):
        torch.cuda.synchronize(self.device)
        self.flat_p.copy_(snap[0]); self.flat_m.copy_(snap[1]); self.flat_v.copy_(snap[2]); self.accum.copy_(snap[3])
        for k, v in snap[4].items():
            self.buffers[k].copy_(v)

    def _record(self, B, weight_decay, wd_overrides, split):
        """Record the launch plan of one step (the launches are executed while being recorded)."""
        rows, cond, snap = self._warm(B, weight_decay, wd_overrides)
        out = dict(rows=rows, cond=cond)
        self._rec = []
        self._dp_dry = True            # the launches run while they are recorded; the replays gather for real
        try:
            self._step_and_adam(rows, cond, B, weight_decay, wd_overrides, split)
            out["fwd"] = self._rec     # (split: the gradient all-reduce sits between the tail and Adam, _defer_tail)
        finally:
            self._rec = None
            self._dp_dry = False
        self._restore(snap)
        return out

    def _capture(self, B, weight_decay, wd_overrides, split):
        """Capture the step into HIP graphs."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            rows, cond, snap = self._warm(B, weight_decay, wd_overrides)
        torch.cuda.current_stream(self.device).wait_stream(s)
        out = dict(rows=rows, cond=cond)
        gf = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf):
            self._step_and_adam(rows, cond, B, weight_decay, wd_overrides, split)
        out["fwd"] = gf
        self._restore(snap)
        return out

    def begin_step(self, alpha_kl, lr, eps=0.01, clip=0.0, stage=None, stage_next=None):
        """Per-step scalars -> device (`hyper`), Philox step counter += 1: one tiny kernel, no host staging.
        `stage` = (rows_src, rows_dst, cond_src, cond_dst, B) also copies the minibatch into the static buffers of
        a recorded plan / graph in the same launch."""
        self.adam_t += 1
        t = self.adam_t
        h = self._hyper_host
        h[0], h[1], h[2], h[3], h[4], h[5] = alpha_kl, lr, 1.0 - 0.9 ** t, math.sqrt(1.0 - 0.999 ** t), eps, clip
        st = torch.cuda.current_stream(self.device).cuda_stream
        if stage is not None:
            rs, rd, cs, cd, B = stage
            ns, nd, Bn = stage_next if stage_next is not None else (None, None, 0)
            _lib.check(self.lib._lib.sca_begin_step_batch_next(
                self.hyper.data_ptr(), self.step_ctr.data_ptr(), float(alpha_kl), float(lr), float(h[2]),
                float(h[3]), float(eps), float(clip), rs.data_ptr(), rd.data_ptr(), cs.data_ptr(), cd.data_ptr(),
                int(B), _lib.ptr(ns), _lib.ptr(nd), int(Bn), st), "begin_step_batch")
            return
        _lib.check(self.lib._lib.sca_begin_step(self.hyper.data_ptr(), self.step_ctr.data_ptr(), float(alpha_kl),
                                                float(lr), float(h[2]), float(h[3]), float(eps), float(clip), st),
                   "begin_step")

    # ------------------------------------------------------------------------------------------ inference
    def encode(self, row_ids, cond, B, mean=True, x_dense=None, eps=None, covs=None):
        """Eval-mode encoder pass: returns (z or mean [B,Z], mean, var) device tensors (views of the workspace)."""
        L, P = self.lib, self.p
        self._pf = None         # ws["x"] is about to be overwritten
        self._ensure(B)
        eps = self._eps_or_zero(eps, B)
        ws = self.ws
        st = torch.cuda.current_stream(self.device).cuda_stream
        ptr = _lib.ptr
        G, C, Z = self.G, self.C, self.Zt
        log1p = 0 if self.recon == "mse" else 1
        if x_dense is not None:
            _lib.check(L.sca_dense_gather_rows(ptr(x_dense), x_dense.stride(0), ptr(row_ids), B, G, log1p,
                                               ws["x"].data_ptr(), G, ws["sf"].data_ptr(), st), "dense_gather")
        else:
            ip, ix, dv = self.csr
            _lib.check(L.sca_csr_gather_rows(ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), ptr(row_ids), B, G, log1p,
                                             ws["x"].data_ptr(), G, ws["sf"].data_ptr(), st), "csr_gather")
        a = ws["x"].data_ptr()
        cond_p = ptr(cond) if C > 0 else None
        if self.use_fused_mid and not self.cond_embed and os.environ.get("SCA_ENCODE_FUSED", "1") == "1":
            # first-layer GEMM, then ONE launch for its epilogue, the remaining encoder layers, the heads and the latent
            # block (sca_mid_fwd with n_dec = 0) instead of a GEMM + epilogue pair per layer
            ly = self.enc[0]
            pre0 = self._gemm(0, 0, a, ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out, ly.n_in, st,
                              max_split=self._mid_max_split)
            self.step_ctr += 1          # fresh noise for every sampled pass
            _lib.check(L.sca_mid_fwd(self._mid_args(B, cond_p, 0.0, eps, {}, pre0=pre0, sample=0 if mean else 1,
                                                    encoder_only=True), st), "mid_fwd[encode]")
            mu = ws["mu_lv"][:B, :Z]
            var = torch.exp(ws["mu_lv"][:B, Z:]) + 1e-4
            return ws["z"][:B], mu, var
        if self.cond_embed:
            self._embed_lookup(covs, B, st)
            cw, cwp, cvec_p, cE = 0, None, ws["cvec"].data_ptr(), self.E
        else:
            cw, cwp, cvec_p, cE = C, cond_p, None, 0
        for ly in self.enc:
            sp, S, stride = self._gemm(0, 0, a, ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out, ly.n_in, st)
            a = self._layer_block_fwd(ly, sp, S, stride, B, (cw, cwp, cvec_p, cE), 0.0, None, False, None, st)
        Hl = self.H[-1]
        sp, S, stride = self._gemm(0, 0, a, Hl, self.heads_w.data_ptr(), Hl, B, 2 * Z, Hl, st)
        self.step_ctr += 1          # fresh noise for every sampled pass
        _lib.check(L.sca_latent_fwd(sp, S, stride, self.heads_b.data_ptr(), B, Z, ptr(eps), self.seed, 2,
                                    self.step_ctr.data_ptr(), 0 if mean else 1, ws["mu_lv"].data_ptr(),
                                    ws["eps"].data_ptr(), ws["z"].data_ptr(), ws["kl_row"].data_ptr(), st),
                   "latent_fwd")
        mu = ws["mu_lv"][:B, :Z]
        var = torch.exp(ws["mu_lv"][:B, Z:]) + 1e-4
        return ws["z"][:B], mu, var

    def decode_first_layer(self, cond, B):
        """y1 = first decoder block applied to the current ws['z'] (eval mode): get_y()."""
        L, P, ws = self.lib, self.p, self.ws
        st = torch.cuda.current_stream(self.device).cuda_stream
        ly = self.dec[0]
        ptr = _lib.ptr
        sp, S, stride = self._gemm(0, 0, ws["z"].data_ptr(), ly.n_in, P[ly.w].data_ptr(), ly.n_in, B, ly.n_out,
                                   ly.n_in, st)
        self._layer_block_fwd(ly, sp, S, stride, B, (self.C, ptr(cond) if self.C > 0 else None, None, 0), 0.0, None,
                              False, None, st)
        return ws[ly.tag + ".h"][:B]
