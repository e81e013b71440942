"""oracle/retrieval_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference retrieval path (reference: evaluate_retrieval.py:22-73).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (semantic-embeddings_amd/) must never do so.

Two restatements live here:

* ``pairwise_retrieval_numpy`` -- the literal NumPy op sequence of the reference
  (evaluate_retrieval.py:43-73) with numexpr's ``A + B - 2 * C`` spelled in float32 NumPy
  (numexpr is not installed; see SURVEY.md section 7 hard part 1d: that boundary is "parity
  unpinned") and ``np.argsort(kind='stable')`` instead of the reference's unstable default, so
  that ties have the canonical (distance, index) order.  This is what bench.py times as the
  ``cpu_baseline`` ("port").

* ``canon_*`` -- ctypes bindings of oracle/canon.c, the explicit-rounding C restatement
  (sequential fp32 FMA chain, NumPy pairwise row norms, canonical ranking).  This is what the GPU
  kernels are compared with bit-for-bit.

Parity status: pinned.  tests/golden/retrieval_*.npz hold rankings produced by the *imported*
reference ``evaluate_retrieval.pairwise_retrieval`` (oracle/make_golden.py); tests/test_oracle.py
checks both restatements against them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile oracle/canon.c -> oracle/libcanon.so (gcc)."""
    so = os.path.join(_HERE, "libcanon.so")
    src = os.path.join(_HERE, "canon.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcanon.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libcanon.so")
        if not os.path.exists(so):
            build()
        lib = ctypes.CDLL(so)
        L = ctypes.c_long
        lib.canon_row_sqsum.argtypes = [_f32p, L, L, _f32p]
        lib.canon_normalize_rows.argtypes = [_f32p, L, L]
        lib.canon_pdist.argtypes = [_f32p, _f32p, ctypes.c_void_p, ctypes.c_void_p, L, L, L, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_int, _f32p, L]
        lib.canon_rank_rows.argtypes = [_f32p, L, L, L, _i32p]
        lib.canon_topk_rows.argtypes = [_f32p, L, L, L, L, ctypes.c_int, _f32p, _i32p]
        lib.canon_topk_merge.argtypes = [_f32p, _i32p, ctypes.c_int, L, ctypes.c_int, _f32p, _i32p]
        for f in (lib.canon_row_sqsum, lib.canon_normalize_rows, lib.canon_pdist, lib.canon_rank_rows,
                  lib.canon_topk_rows, lib.canon_topk_merge):
            f.restype = None
        _LIB = lib
    return _LIB


# --------------------------------------------------------------------------------------------------
# canonical (explicit-rounding) restatement
# --------------------------------------------------------------------------------------------------

def canon_row_sqsum(x):
    """float32 ``np.sum(x ** 2, axis=-1)`` (reference: evaluate_retrieval.py:61)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[0], dtype=np.float32)
    _lib().canon_row_sqsum(x, x.shape[0], x.shape[1], out)
    return out


def canon_normalize_rows(x):
    """float32 ``x / np.linalg.norm(x, axis=-1, keepdims=True)`` (reference: evaluate_retrieval.py:58)."""
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    _lib().canon_normalize_rows(x, x.shape[0], x.shape[1])
    return x


METRIC_COSINE, METRIC_EUCLID, METRIC_DOT = 0, 1, 2


def canon_pdist(a, b=None, metric=METRIC_COSINE, kblocks=None):
    """All-pairs distance matrix with the canonical arithmetic (reference: evaluate_retrieval.py:59,62).

    ``a``/``b`` are used as given (for the cosine branch pass already-normalised rows)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float32)
    q, d = a.shape
    n = b.shape[0]
    out = np.empty((q, n), dtype=np.float32)
    sqa = sqb = None
    pa = pb = None
    if metric == METRIC_EUCLID:
        sqa = canon_row_sqsum(a)
        sqb = sqa if b is a else canon_row_sqsum(b)
        pa, pb = sqa.ctypes.data, sqb.ctypes.data
    kb = None
    nkb = 0
    if kblocks is not None:
        kb = np.ascontiguousarray(kblocks, dtype=np.int32)
        assert int(kb.sum()) == d
        nkb = len(kb)
    _lib().canon_pdist(a, b, pa, pb, q, n, d, int(metric), None if kb is None else kb.ctypes.data, nkb, out, n)
    return out


def canon_rank_rows(pdist):
    """Canonical ``np.argsort(pdist, axis=-1)``: ascending (distance, index); NaN last; -0 == +0."""
    pdist = np.ascontiguousarray(pdist, dtype=np.float32)
    q, n = pdist.shape
    rank = np.empty((q, n), dtype=np.int32)
    _lib().canon_rank_rows(pdist, q, n, n, rank)
    return rank


def canon_topk_rows(pdist, k, col_offset=0):
    pdist = np.ascontiguousarray(pdist, dtype=np.float32)
    q, n = pdist.shape
    od = np.empty((q, k), dtype=np.float32)
    oi = np.empty((q, k), dtype=np.int32)
    _lib().canon_topk_rows(pdist, q, n, n, col_offset, k, od, oi)
    return od, oi


def canon_topk_merge(d, idx):
    """Merge per-shard top-k lists ``[parts, q, k]`` into the global canonical top-k ``[q, k]``."""
    d = np.ascontiguousarray(d, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    parts, q, k = d.shape
    od = np.empty((q, k), dtype=np.float32)
    oi = np.empty((q, k), dtype=np.int32)
    _lib().canon_topk_merge(d, idx, parts, q, k, od, oi)
    return od, oi


def canon_retrieval(features, normalize=False, kblocks=None):
    """Full canonical path: (normalise) -> distance -> ranking.  Returns (pdist f32 [N,N], rank i32 [N,N])."""
    f = np.array(features, dtype=np.float32, order="C", copy=True)
    if normalize:
        f = canon_normalize_rows(f)
        pd = canon_pdist(f, None, METRIC_COSINE, kblocks)
    else:
        pd = canon_pdist(f, None, METRIC_EUCLID, kblocks)
    return pd, canon_rank_rows(pd)


# --------------------------------------------------------------------------------------------------
# literal NumPy restatement of the reference op sequence (this is the timed CPU baseline)
# --------------------------------------------------------------------------------------------------

def pdist_numpy(features, normalize=False, queries=None):
    """The reference's distance matrix (evaluate_retrieval.py:57-63) on float32 NumPy/BLAS.

    ``features`` is modified in place when ``normalize`` is set, like the reference does.
    ``queries`` (optional row slice) restricts the left operand -- used only to bound the timed
    CPU sample; the reference always uses all rows."""
    if normalize:
        features /= np.linalg.norm(features, axis=-1, keepdims=True)
        lhs = features if queries is None else features[queries]
        return -np.dot(lhs, features.T)
    sqnorm = np.sum(features ** 2, axis=-1)
    lhs = features if queries is None else features[queries]
    sq_l = sqnorm if queries is None else sqnorm[queries]
    c = np.dot(lhs, features.T)
    # numexpr 'A + B - 2 * C' == (A + B) - (2 * C), float32 throughout
    return (sq_l[:, None] + sqnorm[None, :]) - np.float32(2) * c


def pairwise_retrieval_numpy(features, normalize=False, queries=None):
    """Distance + ranking as the reference computes them, ties in canonical order."""
    pd = pdist_numpy(features, normalize, queries)
    return np.argsort(pd, axis=-1, kind="stable")


def probe_host_blas_is_fma_chain(d=100, n=256, seed=0):
    """True when this host's BLAS sgemm equals the sequential fp32 FMA chain for depth ``d``."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return bool(np.array_equal(np.dot(x, x.T), canon_pdist(x, None, METRIC_DOT)))


# --------------------------------------------------------------------------------------------------
# topk_head_*.npz fixtures (oracle/make_golden.py topk_goldens): features rebuilt from their seed
# --------------------------------------------------------------------------------------------------

def topk_feature_matrix(kind, seed, n, emb=None):
    """The feature matrices of the topk_head_* fixtures (they store seed / SHA-1 instead of 2.5 MB of incompressible
    floats).  'inet': ILSVRC-like trained features (class-embedding row + noise, D = 1000); 'gauss': plain gaussian, D = 555."""
    rng = np.random.default_rng(seed)
    if kind == "inet":
        y = rng.integers(0, emb.shape[0], size=n)
        return (emb[y] + 0.03 * rng.standard_normal((n, emb.shape[1]))).astype(np.float32)
    return rng.standard_normal((n, 555)).astype(np.float32)


def load_topk_fixture(path):
    """-> (features f32 [n, D], normalize, kblocks list, ref_head int64 [n, 256]) of a topk_head_*.npz fixture;
    raises if this NumPy does not regenerate the exact feature bytes the reference was run on."""
    import hashlib
    g = np.load(path)
    emb = np.load(os.path.join(os.path.dirname(path), "imagenet_mintree_unitsphere.npz"))["embedding"]
    feat = topk_feature_matrix(str(g["kind"]), int(g["seed"]), int(g["n"]), emb)
    if hashlib.sha1(feat.tobytes()).hexdigest() != str(g["sha1"]):
        raise RuntimeError("fixture %s: regenerated features differ from the ones the reference ranked" % path)
    return feat, bool(g["normalize"]), g["kblocks"].tolist(), g["ref_head"].astype(np.int64)
