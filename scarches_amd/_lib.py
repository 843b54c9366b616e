"""ctypes binding of libscarches_hip.so (C ABI declared in include/scarches_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libscarches_hip.so")

P = C.c_void_p
I = C.c_int
LL = C.c_longlong
F = C.c_float
U64 = C.c_uint64
U32 = C.c_uint32

class TailJob(C.Structure):
    """sca_tail_job of include/scarches_hip.h."""
    _fields_ = [("kind", C.c_int32), ("S", C.c_int32), ("stride", C.c_longlong), ("src", C.c_void_p),
                ("dst", C.c_void_p), ("n", C.c_longlong), ("dWc", C.c_void_p), ("cond", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("pad", C.c_int32)]


class GemmJob(C.Structure):
    """sca_gemm_job of include/scarches_hip.h."""
    _fields_ = [("amode", C.c_int32), ("bmode", C.c_int32), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("splitk", C.c_int32), ("pad", C.c_int32), ("slab_stride", C.c_longlong)]


class MidLayer(C.Structure):
    """sca_mid_layer of include/scarches_hip.h."""
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("Wc", C.c_void_p), ("mask", C.c_void_p),
                ("yn", C.c_void_p), ("rs", C.c_void_p), ("h", C.c_void_p), ("da", C.c_void_p),
                ("N", C.c_int32), ("K", C.c_int32), ("rng_stream", C.c_uint32), ("pad", C.c_int32)]


class MidArgs(C.Structure):
    """sca_mid_args of include/scarches_hip.h."""
    _fields_ = [("enc", MidLayer * 4), ("dec", MidLayer * 4), ("n_enc", C.c_int32), ("n_dec", C.c_int32),
                ("Wh", C.c_void_p), ("bh", C.c_void_p), ("eps_in", C.c_void_p), ("mu_lv", C.c_void_p),
                ("eps_out", C.c_void_p), ("z", C.c_void_p), ("kl_row", C.c_void_p),
                ("Z", C.c_int32), ("Kh", C.c_int32), ("rng_latent", C.c_uint32), ("sample", C.c_int32),
                ("B", C.c_int32), ("C", C.c_int32), ("cond", C.c_void_p),
                ("norm", C.c_int32), ("p", C.c_float), ("seed", C.c_uint64), ("step", C.c_void_p),
                ("h_in", C.c_void_p), ("K0", C.c_int32), ("S", C.c_int32),
                ("dh", C.c_void_p), ("dh_stride", C.c_longlong), ("dz2", C.c_void_p), ("hyper", C.c_void_p),
                ("dheads", C.c_void_p), ("dh_out", C.c_void_p),
                ("fuse_first", C.c_int32), ("S0", C.c_int32), ("pre0", C.c_void_p), ("pre0_stride", C.c_longlong),
                ("first", MidLayer)]


# name -> argtypes (restype is always int); mirrors include/scarches_hip.h one to one
SIGNATURES = {
    "sca_abi_version": [],
    "sca_device_name": [C.c_char_p, I],
    "sca_csr_gather_rows": [P, P, P, P, I, I, I, P, I, P, P],
    "sca_dense_gather_rows": [P, I, P, I, I, I, P, I, P, P],
    "sca_gemm_splitk_heuristic": [I, I, I],
    "sca_gemm_f32": [I, I, P, I, P, I, P, I, I, I, I, I, LL, P, P],
    "sca_gemm_f32_cfg": [I, I, P, I, P, I, P, I, I, I, I, I, LL, P, I, I, P],
    "sca_splitk_reduce": [P, I, LL, P, LL, P],
    "sca_gemm_kk_splitk_heuristic": [I, I, I],
    "sca_gemm_kk_f32": [P, I, P, I, P, I, I, I, I, I, LL, P],
    "sca_gemm_kk_supported": [P, I, P, I, I, I],
    "sca_gemm_group_max_jobs": [],
    "sca_gemm_group_f32": [P, I, P],
    "sca_layer_fwd": [P, I, LL, I, I, P, P, I, P, P, I, I, I, F, P, U64, U32, P, P, P, P, P],
    "sca_layer_bwd": [P, I, LL, P, I, I, P, P, I, I, F, P, U64, U32, P, P, P],
    "sca_cond_colsum": [P, I, I, P, I, P, P, P, P],
    "sca_latent_fwd": [P, I, LL, P, I, I, P, U64, U32, P, I, P, P, P, P, P],
    "sca_latent_bwd": [P, I, LL, P, I, LL, P, P, P, I, I, P, P],
    "sca_theta_prep": [P, I, I, P, P, P],
    "sca_theta_prep_zero": [P, I, I, P, P, P, LL, P, LL, P],
    "sca_recon_rows": [I, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, I, P, P, P],
    "sca_out_nb_fused_supported": [I, I],
    "sca_nb_gamma_table": [P, I, I, P, P],
    "sca_out_nb_fused_debug_stamps": [P],
    "sca_out_nb_fused": [P, I, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, P, I, P, I, P, P, P, I, P],
    "sca_recon_rows_tab": [I, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, I, P, P, P, P],
    "sca_poisson_rows": [P, I, P, P, P, P, P, I, I, I, P, I, P],
    "sca_nb_elem_rows": [I, P, I, P, P, P, P, P, P, P, I, I, I, I, P, P, I, P, P],
    "sca_mse_rows": [P, I, P, I, I, I, I, P, P, I, I, P, P, P],
    "sca_transpose_scale": [P, I, I, F, P, P],
    "sca_exp_f32": [P, P, LL, P],
    "sca_mul_f32": [P, P, P, LL, P],
    "sca_mmd_num_slabs": [I, I],
    "sca_mmd_num_partials": [I, I],
    "sca_mmd_fwd_bwd": [P, I, P, I, I, I, I, F, P, P, P, P, LL, P],
    "sca_mmd_select_form": [I],
    "sca_mmd_reserve_lds": [I],
    "sca_mmd_prepare": [P, I, I, I, F, P, P, P],
    "sca_mmd_prepared_supported": [I, I],
    "sca_mmd_fwd_bwd_prepared": [P, I, P, I, I, I, P, I, I, P, P, LL, P],
    "sca_mmd_rows_num_slabs": [I, I, I],
    "sca_mmd_rows_num_partials": [I, I, I],
    "sca_mmd_rows_fwd_bwd": [P, I, P, I, I, I, I, F, P, P, I, I, P, P, LL, P],
    "sca_adam_step": [P, P, P, P, LL, P, F, F, F, P],
    "sca_finalize_losses": [P, F, P, I, P, I, P, P, P, P],
    "sca_begin_step": [P, P, F, F, F, F, F, F, P],
    "sca_begin_step_batch": [P, P, F, F, F, F, F, F, P, P, P, P, I, P],
    "sca_begin_step_batch_next": [P, P, F, F, F, F, F, F, P, P, P, P, I, P, P, I, P],
    "sca_fill_zero": [P, LL, P],
    "sca_masked_copy": [P, I, P, I, P, I, I, I, P],
    "sca_masked_max_cols": [],
    "sca_masked_max_width": [],
    "sca_masked_fwd_tiles_per_wg": [I, I],
    "sca_masked_dx_slabs": [I, I],
    "sca_masked_dw_slabs": [I, I],
    "sca_masked_pack": [P, I, P, P, I, I, I, I, P, P],
    "sca_masked_unpack": [P, I, LL, P, P, I, I, I, I, P, I, P],
    "sca_masked_fwd": [P, I, P, P, P, I, I, I, I, I, P, I, P],
    "sca_masked_dx": [P, I, P, P, P, I, I, I, I, I, P, LL, P],
    "sca_masked_dw": [P, I, P, I, P, P, I, I, I, I, I, P, LL, P],
    "sca_prox_l1": [P, LL, F, P, P],
    "sca_prox_group_lasso": [P, I, I, F, P, P],
    "sca_prox_group_lasso_ws": [P, I, I, F, P, P, P],
    "sca_hsic_workspace_floats": [I],
    "sca_hsic_fwd_bwd": [P, I, I, I, P, F, F, F, P, P, P, I, P, I, P],
    "sca_mid_max_width": [],
    "sca_mid_fwd": [P, P],
    "sca_mid_bwd": [P, P],
    "sca_mid_fwd_staged": [P, P],
    "sca_mid_bwd_staged": [P, P],
    "sca_knn_max_k": [],
    "sca_knn_num_chunks": [I, I],
    "sca_knn_topk": [P, I, P, I, I, I, P, P, P, P, P],
    "sca_knn_vote": [P, P, I, I, P, P, P, P],
    "sca_knn_static_max_entries": [],
    "sca_knn_topk_static": [P, I, P, I, I, I, P, P, P],
    "sca_grad_tail_max_jobs": [],
    "sca_grad_tail": [P, I, P],
    "sca_bn_fwd": [P, I, I, P, P, P, P, I, I, F, F, P, U64, U32, P, P, P, P, P],
    "sca_bn_bwd": [P, I, LL, P, I, I, P, P, P, P, I, F, P, U64, U32, P, P, P, P, P, P],
    "sca_embed_lookup": [P, I, I, P, I, I, F, P, P, I, P],
    "sca_embed_scatter_add": [P, P, I, I, I, I, P, I, P, P],
    "sca_proto_workspace_bytes": [I, I, I],
    "sca_proto_loss": [P, I, I, I, P, I, P, I, P, F, P, I, F, P, P, P, I, P, I, P, P],
    "sca_proto_classify": [P, I, I, I, P, I, P, I, I, P, P, P, P],
    "sca_class_moments": [P, I, I, I, P, I, P, I, P, P, P, P],
    "sca_class_moments_finish": [P, P, P, I, I, P, P, P, P, P, P],
}
RESTYPES = {"sca_hsic_workspace_floats": LL, "sca_proto_workspace_bytes": LL, "sca_knn_static_max_entries": LL}

_lib = None
ABI_VERSION = 3       # sca_abi_version() of the library this binding was written against


class HipLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles; no GPU needed)."""
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose:
        print(res.stdout[-2000:], res.stderr[-2000:])
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise HipLibraryError("building libscarches_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return LIB_PATH


def load():
    """Load the shared library and declare every symbol of the header.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found - build it with `make -C scarches_amd/csrc` (or __graft_entry__.build()); "
            "scarches_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, I)
    got = lib.sca_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it "
                              "with `make -C scarches_amd/csrc`")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise HipLibraryError(f"{what} failed with code {rc}"
                              + (" (unsupported shape/argument)" if rc < 0 else " (hipError_t)"))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
