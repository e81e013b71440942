"""ctypes binding of libsehip.so (C ABI declared in include/sehip.h).

The library is built in-tree by ``csrc/Makefile`` (``hipcc --offload-arch=gfx950``).  There is NO
CPU fallback: if the shared object is missing, or a kernel is asked to run without a ROCm device,
the call raises -- the product path never silently degrades to PyTorch/NumPy code.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.environ.get("SEHIP_LIB") or os.path.join(_HERE, "libsehip.so")   # SEHIP_LIB: tuning builds only
TUNING_LIB_PATH = os.path.join(_HERE, "libsehip_tuning.so")   # -DSE_TUNING build: honours the SE_* variant / profile switches

SE_OK = 0
DTYPE_F32, DTYPE_BF16 = 0, 1
METRIC_COSINE, METRIC_EUCLID, METRIC_DOT = 0, 1, 2
TOPK_MAX = 2048

# every symbol include/sehip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "se_version", "se_last_error", "se_build_arch", "se_phase_timing", "se_phase_timing_read",
    "se_cosine_loss_fwd", "se_cosine_loss_bwd", "se_sqdist_loss_fwd", "se_sqdist_loss_bwd", "se_l2norm_fwd", "se_l2norm_bwd", "se_nn_accuracy_workspace_bytes", "se_nn_accuracy",
    "se_labelembed_aux_floats", "se_labelembed_loss_fwd", "se_labelembed_loss_bwd",
    "se_devise_aux_floats", "se_devise_loss_fwd", "se_devise_loss_bwd",
    "se_row_sqnorm", "se_normalize_rows", "se_pairwise_dist",
    "se_rank_rows_workspace_bytes", "se_rank_rows", "se_rank_rows_init_workspace_bytes", "se_rank_rows_init", "se_rank_rows_check_workspace_bytes", "se_rank_rows_check",
    "se_topk_rows", "se_topk_merge", "se_topk_merge_packed",
    "se_retrieve_topk_workspace_bytes", "se_retrieve_topk", "se_hierarchical_precision", "se_hierarchical_precision_r16",
    "se_hprec_order_workspace_bytes", "se_hprec_curve_len", "se_hprec_reciprocal_curves",
)


class SehipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into sehip/libsehip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "sehip.h"))
    built = [os.path.join(_HERE, "libsehip.so"), TUNING_LIB_PATH]
    stale = any((not os.path.exists(b)) or any(os.path.getmtime(s) > os.path.getmtime(b) for s in srcs if os.path.exists(s))
                for b in built)
    if force or stale:
        cmd = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or out.returncode != 0:
            print(out.stdout)
        if out.returncode != 0:
            raise SehipError("building libsehip.so failed (see output above)")
    return LIB_PATH


_lib = None


def lib():
    """Load libsehip.so (after torch, so both share one HIP runtime) and declare signatures."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- loads libamdhip64.so.7 first; libsehip.so then binds to the same runtime
    if not os.path.exists(LIB_PATH):
        raise SehipError(
            "libsehip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C semantic-embeddings_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_i64, c_int, c_f, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p
    L.se_version.restype = c_int
    L.se_last_error.restype = ctypes.c_char_p
    L.se_build_arch.restype = ctypes.c_char_p
    L.se_phase_timing.argtypes = [c_int]
    L.se_phase_timing_read.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_f), c_int, ctypes.POINTER(c_i64)]
    L.se_cosine_loss_fwd.argtypes = [vp, c_int, c_i64, vp, vp, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp]
    L.se_cosine_loss_bwd.argtypes = [vp, c_int, c_i64, vp, vp, c_i64, vp, c_f, c_i64, c_i64, c_i64, vp, c_int, c_i64, vp]
    L.se_sqdist_loss_fwd.argtypes = [vp, c_int, c_i64, vp, vp, c_i64, c_i64, c_i64, c_i64, vp, vp, vp, vp]
    L.se_sqdist_loss_bwd.argtypes = [vp, c_int, c_i64, vp, vp, c_i64, vp, c_f, c_i64, c_i64, c_i64, vp, c_int, c_i64, vp]
    L.se_l2norm_fwd.argtypes = [vp, c_int, c_i64, c_i64, c_i64, vp, c_i64, vp, vp]
    L.se_l2norm_bwd.argtypes = [vp, c_i64, vp, c_i64, vp, c_i64, c_i64, vp, c_i64, vp]
    L.se_nn_accuracy_workspace_bytes.argtypes = [c_i64, c_i64]
    L.se_nn_accuracy_workspace_bytes.restype = c_i64
    L.se_nn_accuracy.argtypes = [vp, c_i64, vp, vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, vp, vp, c_i64, vp, vp, c_i64, vp]
    L.se_labelembed_aux_floats.argtypes = [c_i64]
    L.se_labelembed_aux_floats.restype = c_i64
    L.se_labelembed_loss_fwd.argtypes = [vp, c_i64, vp, c_i64, vp, c_i64, vp, c_i64, c_i64, c_f, c_f, c_f, vp, vp, vp]
    L.se_labelembed_loss_bwd.argtypes = [vp, c_i64, vp, c_i64, vp, c_i64, vp, vp, c_f, c_i64, c_i64, c_f, c_f, c_f, vp,
                                         vp, c_i64, vp, c_i64, vp, c_i64, vp]
    L.se_devise_aux_floats.argtypes = [c_i64, c_i64]
    L.se_devise_aux_floats.restype = c_i64
    L.se_devise_loss_fwd.argtypes = [vp, c_i64, vp, vp, c_i64, vp, c_i64, c_i64, c_i64, c_i64, c_f, vp, vp, vp]
    L.se_devise_loss_bwd.argtypes = [vp, vp, c_i64, vp, c_i64, vp, c_f, c_i64, c_i64, c_i64, vp, vp, c_i64, vp]
    L.se_row_sqnorm.argtypes = [vp, c_i64, c_i64, c_i64, vp, vp]
    L.se_normalize_rows.argtypes = [vp, c_i64, c_i64, c_i64, vp]
    L.se_pairwise_dist.argtypes = [vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i64, c_i64, c_int,
                                   ctypes.POINTER(ctypes.c_int32), c_int, vp, c_i64, vp]
    L.se_hierarchical_precision.argtypes = [vp, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, c_int, vp, c_i64, vp, c_int,
                                            c_i64, c_int, vp, c_i64, vp, vp]
    L.se_hierarchical_precision_r16.argtypes = L.se_hierarchical_precision.argtypes
    L.se_hprec_order_workspace_bytes.argtypes = [c_i64]
    L.se_hprec_order_workspace_bytes.restype = c_i64
    L.se_hprec_curve_len.argtypes = [c_i64]
    L.se_hprec_curve_len.restype = c_i64
    L.se_hprec_reciprocal_curves.argtypes = [vp, vp, c_i64, c_int, c_i64, vp, vp]
    L.se_rank_rows_workspace_bytes.argtypes = [c_i64, c_i64]
    L.se_rank_rows_workspace_bytes.restype = c_i64
    L.se_rank_rows.argtypes = [vp, c_i64, c_i64, c_i64, vp, c_int, c_i64, vp, c_i64, vp]
    L.se_rank_rows_init_workspace_bytes.argtypes = []
    L.se_rank_rows_init_workspace_bytes.restype = c_i64
    L.se_rank_rows_init.argtypes = [vp, c_i64, vp]
    L.se_rank_rows_check_workspace_bytes.argtypes = []
    L.se_rank_rows_check_workspace_bytes.restype = c_i64
    L.se_rank_rows_check.argtypes = [vp, c_i64, c_i64, c_i64, vp, c_int, c_i64, vp, c_i64, ctypes.POINTER(c_i64), vp]
    L.se_topk_rows.argtypes = [vp, c_i64, c_i64, c_i64, c_i64, c_int, vp, vp, vp]
    L.se_topk_merge.argtypes = [vp, vp, c_int, c_i64, c_int, vp, vp, vp]
    L.se_topk_merge_packed.argtypes = [vp, c_int, c_i64, c_int, vp, vp, vp]
    L.se_retrieve_topk_workspace_bytes.argtypes = [c_i64, c_i64, c_i64, c_i64, c_int]
    L.se_retrieve_topk_workspace_bytes.restype = c_i64
    L.se_retrieve_topk.argtypes = [vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i64, c_i64, c_int,
                                   ctypes.POINTER(ctypes.c_int32), c_int, c_i64, c_int, vp, vp, vp, c_i64, vp]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here means the .so is stale w.r.t. include/sehip.h
    _lib = L
    return L


def check(rc, what):
    if rc != SE_OK:
        msg = lib().se_last_error()
        raise SehipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))


def require_gpu(*tensors):
    """The product path refuses to run anywhere but on a ROCm device."""
    import torch
    if not torch.cuda.is_available():
        raise SehipError("sehip kernels need a ROCm GPU (torch.cuda.is_available() is False); "
                         "there is no CPU fallback")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SehipError("sehip kernels take device tensors; got a %s tensor" % t.device)


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
