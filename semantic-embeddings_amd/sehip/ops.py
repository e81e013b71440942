"""Torch-facing wrappers of the HIP hot path (device tensors in, device tensors out).

Every function here calls straight through the C ABI of libsehip.so on the current torch stream;
PyTorch only provides device memory, streams and autograd plumbing.  Reference citations are into
the cvjena/semantic-embeddings checkout.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (DTYPE_BF16, DTYPE_F32, METRIC_COSINE, METRIC_DOT, METRIC_EUCLID, SehipError, check, lib, ptr,
                   require_gpu, stream_ptr)

__all__ = [
    "cosine_embedding_loss", "cosine_loss_forward", "cosine_loss_backward", "l2norm", "nn_accuracy", "labelembed_loss",
    "devise_ranking_loss",
    "row_sqnorm", "normalize_rows_", "pairwise_dist", "rank_rows", "rank_rows_check", "topk_rows", "topk_merge", "retrieve_topk",
    "hierarchical_precision", "hprec_reciprocal_curves",
    "METRIC_COSINE", "METRIC_EUCLID", "METRIC_DOT",
]


def _dtype_code(t):
    if t.dtype == torch.float32:
        return DTYPE_F32
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    raise SehipError("features must be float32 or bfloat16, got %s" % t.dtype)


def _rows(t, what):
    if t.dim() != 2 or t.stride(1) != 1:
        raise SehipError("%s must be a 2-d tensor with contiguous rows" % what)
    return t


# --------------------------------------------------------------------------------------------
# training side
# --------------------------------------------------------------------------------------------

def cosine_loss_forward(x, labels, embedding, want_xhat=True):
    """Fused l2norm + gather + inv_correlation (+ batch mean).

    reference: utils.l2norm (utils.py:125-127), transform_inputs (learn_image_embeddings.py:48-50),
    utils.inv_correlation (utils.py:44-46).  Returns (xhat | None, inv_norm, loss_i, loss_mean)."""
    require_gpu(x, labels, embedding)
    _rows(x, "x"); _rows(embedding, "embedding")
    B, D = x.shape
    C = embedding.shape[0]
    if embedding.shape[1] != D or embedding.dtype != torch.float32:
        raise SehipError("embedding must be float32 [C, %d]" % D)
    if labels.dtype != torch.int64 or labels.numel() != B or not labels.is_contiguous():
        raise SehipError("labels must be a contiguous int64 [B] tensor")
    xhat = torch.empty((B, D), dtype=torch.float32, device=x.device) if want_xhat else None
    inv_norm = torch.empty((B,), dtype=torch.float32, device=x.device)
    loss_i = torch.empty((B,), dtype=torch.float32, device=x.device)
    loss_mean = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib().se_cosine_loss_fwd(ptr(x), _dtype_code(x), x.stride(0), ptr(labels), ptr(embedding),
                                   embedding.stride(0), B, D, C, ptr(xhat), D, ptr(inv_norm), ptr(loss_i),
                                   ptr(loss_mean), stream_ptr()), "se_cosine_loss_fwd")
    return xhat, inv_norm, loss_i, loss_mean


def cosine_loss_backward(x, labels, embedding, grad_loss_i=None, grad_scale=1.0, out_dtype=None):
    """Closed-form backward of cosine_loss_forward w.r.t. x (what TF autodiff derives from
    utils.py:44-46,125-127)."""
    require_gpu(x, labels, embedding, grad_loss_i)
    _rows(x, "x"); _rows(embedding, "embedding")
    B, D = x.shape
    C = embedding.shape[0]
    out_dtype = out_dtype or x.dtype
    dx = torch.empty((B, D), dtype=out_dtype, device=x.device)
    if grad_loss_i is not None:
        grad_loss_i = grad_loss_i.to(torch.float32).contiguous()
    check(lib().se_cosine_loss_bwd(ptr(x), _dtype_code(x), x.stride(0), ptr(labels), ptr(embedding),
                                   embedding.stride(0), ptr(grad_loss_i), ctypes.c_float(grad_scale), B, D, C,
                                   ptr(dx), _dtype_code(dx), D, stream_ptr()), "se_cosine_loss_bwd")
    return dx


def _check_loss_inputs(labels, embedding, what):
    """The loss kernels gather ``embedding[label]`` with the label CLAMPED to [0, C - 1] (a device kernel cannot raise the IndexError
    the reference's ``embedding[y]`` gather would), and they return no gradient for the class-embedding table (it is a precomputed
    constant in the reference: learn_image_embeddings.py:48-50).  ``SEHIP_CHECK_LABELS=1`` validates the labels on the host
    (one synchronising min / max per call: a debugging aid); a table that asks for a gradient is refused always."""
    if embedding.requires_grad:
        raise SehipError("%s: the class-embedding table is a constant of this loss (no gradient is computed for it); detach() it" % what)
    if os.environ.get("SEHIP_CHECK_LABELS"):
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= embedding.shape[0]:
            raise IndexError("%s: labels span [%d, %d] but the embedding table has %d rows" % (what, lo, hi, embedding.shape[0]))


class _CosineEmbeddingLoss(torch.autograd.Function):
    """Per-sample loss_i = 1 - <l2norm(x_i), E[y_i]> with the HIP forward/backward; also returns
    the normalised features the forward kernel produces anyway (non-differentiable by-product)."""

    @staticmethod
    def forward(ctx, x, labels, embedding, want_xhat):
        _check_loss_inputs(labels, embedding, "cosine_embedding_loss")
        x = x if x.stride(-1) == 1 else x.contiguous()
        xhat, _, loss_i, _ = cosine_loss_forward(x, labels, embedding, want_xhat=want_xhat)
        ctx.save_for_backward(x, labels, embedding)
        if xhat is None:
            xhat = loss_i.new_empty(0)
        ctx.mark_non_differentiable(xhat)
        return loss_i, xhat

    @staticmethod
    def backward(ctx, grad_loss_i, _grad_xhat):
        x, labels, embedding = ctx.saved_tensors
        dx = cosine_loss_backward(x, labels, embedding, grad_loss_i.contiguous())
        return dx, None, None, None


def cosine_embedding_loss(x, labels, embedding, reduction="mean", return_normalized=False):
    """Differentiable cosine-embedding loss on un-normalised features ``x`` [B, D].

    Equivalent to the reference's ``Lambda(utils.l2norm)`` head followed by
    ``utils.inv_correlation(embedding[y], .)`` and Keras' batch mean.  With ``return_normalized``
    the L2-normalised features (what the reference's model outputs) are returned as well."""
    loss_i, xhat = _CosineEmbeddingLoss.apply(x, labels, embedding, bool(return_normalized))
    if reduction == "mean":
        loss_i = loss_i.mean()
    elif reduction == "sum":
        loss_i = loss_i.sum()
    return (loss_i, xhat) if return_normalized else loss_i


def sqdist_loss_forward(x, labels, embedding, want_dist=False):
    """``se_sqdist_loss_fwd``: loss_i = sum_d (x - E[y])^2 (utils.squared_distance on transform_inputs' gather, utils.py:34-36,
    learn_image_embeddings.py:48-50) and, on request, dist_i = sqrt(loss_i) (utils.mean_distance, utils.py:39-41).
    Returns (loss_i, dist_i | None, loss_mean)."""
    require_gpu(x, labels, embedding)
    _rows(x, "x"); _rows(embedding, "embedding")
    B, D = x.shape
    C = embedding.shape[0]
    if embedding.shape[1] != D or embedding.dtype != torch.float32:
        raise SehipError("embedding must be float32 [C, %d]" % D)
    if labels.dtype != torch.int64 or labels.numel() != B or not labels.is_contiguous():
        raise SehipError("labels must be a contiguous int64 [B] tensor")
    loss_i = torch.empty((B,), dtype=torch.float32, device=x.device)
    dist_i = torch.empty((B,), dtype=torch.float32, device=x.device) if want_dist else None
    loss_mean = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib().se_sqdist_loss_fwd(ptr(x), _dtype_code(x), x.stride(0), ptr(labels), ptr(embedding), embedding.stride(0), B, D, C,
                                   ptr(loss_i), ptr(dist_i), ptr(loss_mean), stream_ptr()), "se_sqdist_loss_fwd")
    return loss_i, dist_i, loss_mean


def sqdist_loss_backward(x, labels, embedding, grad_loss_i=None, grad_scale=1.0, out_dtype=None):
    """``se_sqdist_loss_bwd``: dx = 2 w (x - E[y])."""
    require_gpu(x, labels, embedding, grad_loss_i)
    _rows(x, "x"); _rows(embedding, "embedding")
    B, D = x.shape
    C = embedding.shape[0]
    dx = torch.empty((B, D), dtype=out_dtype or x.dtype, device=x.device)
    if grad_loss_i is not None:
        grad_loss_i = grad_loss_i.to(torch.float32).contiguous()
    check(lib().se_sqdist_loss_bwd(ptr(x), _dtype_code(x), x.stride(0), ptr(labels), ptr(embedding), embedding.stride(0), ptr(grad_loss_i),
                                   ctypes.c_float(grad_scale), B, D, C, ptr(dx), _dtype_code(dx), D, stream_ptr()), "se_sqdist_loss_bwd")
    return dx


class _SquaredDistanceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, labels, embedding):
        _check_loss_inputs(labels, embedding, "squared_distance_loss")
        x = x if x.stride(-1) == 1 else x.contiguous()
        loss_i, _, _ = sqdist_loss_forward(x, labels, embedding)
        ctx.save_for_backward(x, labels, embedding)
        return loss_i

    @staticmethod
    def backward(ctx, grad_loss_i):
        x, labels, embedding = ctx.saved_tensors
        return sqdist_loss_backward(x, labels, embedding, grad_loss_i.contiguous()), None, None


def squared_distance_loss(x, labels, embedding, reduction="none"):
    """Differentiable ``utils.squared_distance(embedding[labels], x)`` (the `--loss mse` training loss) in one HIP launch forward,
    one backward; ``reduction``: "none" (Keras-style per-sample tensor), "mean" or "sum"."""
    loss_i = _SquaredDistanceLoss.apply(x, labels, embedding)
    if reduction == "mean":
        return loss_i.mean()
    if reduction == "sum":
        return loss_i.sum()
    return loss_i


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        B, D = x2.shape
        xhat = torch.empty((B, D), dtype=torch.float32, device=x.device)
        inv = torch.empty((B,), dtype=torch.float32, device=x.device)
        check(lib().se_l2norm_fwd(ptr(x2), _dtype_code(x2), x2.stride(0), B, D, ptr(xhat), D, ptr(inv), stream_ptr()),
              "se_l2norm_fwd")
        ctx.save_for_backward(xhat, inv)
        ctx.in_dtype = x.dtype
        return xhat.reshape(shape)

    @staticmethod
    def backward(ctx, grad):
        xhat, inv = ctx.saved_tensors
        B, D = xhat.shape
        g = grad.reshape(B, D).to(torch.float32).contiguous()
        dx = torch.empty_like(xhat)
        check(lib().se_l2norm_bwd(ptr(g), D, ptr(xhat), D, ptr(inv), B, D, ptr(dx), D, stream_ptr()), "se_l2norm_bwd")
        return dx.reshape(grad.shape).to(ctx.in_dtype)


def l2norm(x):
    """reference: utils.l2norm (utils.py:125-127) == tf.nn.l2_normalize(x, -1); float32 output."""
    return _L2Norm.apply(x)


def nn_accuracy(y_pred, labels, embedding, dot_prod_sim=False, k=1, want_scores=False, want_best=False):
    """reference: utils.nn_accuracy(embedding, dot_prod_sim, k)(embedding[labels], y_pred) (utils.py:57-100).

    Returns acc [B] float32 (and optionally the [B, C] score matrix and the best class per row)."""
    require_gpu(y_pred, labels, embedding)
    y_pred = _rows(y_pred.to(torch.float32), "y_pred")
    _rows(embedding, "embedding")
    B, D = y_pred.shape
    C = embedding.shape[0]
    acc = torch.empty((B,), dtype=torch.float32, device=y_pred.device)
    scores = torch.empty((B, C), dtype=torch.float32, device=y_pred.device) if want_scores else None
    best = torch.empty((B,), dtype=torch.int32, device=y_pred.device) if want_best else None
    need = int(lib().se_nn_accuracy_workspace_bytes(B, C))
    ws = torch.empty((need // 8,), dtype=torch.int64, device=y_pred.device) if need else None     # (fresh: the call may be captured in a HIP graph)
    check(lib().se_nn_accuracy(ptr(y_pred), y_pred.stride(0), ptr(labels), ptr(embedding), embedding.stride(0),
                               B, D, C, int(bool(dot_prod_sim)), int(k), ptr(acc), ptr(scores), C, ptr(best),
                               ptr(ws), need, stream_ptr()), "se_nn_accuracy")
    out = (acc,)
    if want_scores:
        out += (scores,)
    if want_best:
        out += (best,)
    return out if len(out) > 1 else acc


class _LabelEmbedLoss(torch.autograd.Function):
    """reference: labelembed_loss (learn_labelembedding.py:21-37) and its TF-autodiff backward."""

    @staticmethod
    def forward(ctx, out1, out2, tar, targets, tau, alpha, beta):
        require_gpu(out1, out2, tar, targets)
        for t, name in ((out1, "out1"), (out2, "out2"), (tar, "tar")):
            _f32_rows(t, name)
        B, C = out1.shape
        if out2.shape != (B, C) or tar.shape != (B, C):
            raise SehipError("out1, out2 and tar must share the shape [B, C]")
        if targets.dtype != torch.int64 or targets.numel() != B or not targets.is_contiguous():
            raise SehipError("targets must be a contiguous int64 [B] tensor")
        loss_i = torch.empty((B,), dtype=torch.float32, device=out1.device)
        aux = torch.empty((max(int(lib().se_labelembed_aux_floats(B)), 1),), dtype=torch.float32, device=out1.device)
        check(lib().se_labelembed_loss_fwd(ptr(out1), out1.stride(0), ptr(out2), out2.stride(0), ptr(tar), tar.stride(0),
                                           ptr(targets), B, C, float(tau), float(alpha), float(beta), ptr(loss_i), ptr(aux),
                                           stream_ptr()), "se_labelembed_loss_fwd")
        ctx.save_for_backward(out1, out2, tar, targets, aux)
        ctx.hyper = (float(tau), float(alpha), float(beta))
        return loss_i

    @staticmethod
    def backward(ctx, grad):
        out1, out2, tar, targets, aux = ctx.saved_tensors
        tau, alpha, beta = ctx.hyper
        B, C = out1.shape
        grad = grad.contiguous().to(torch.float32)
        need = ctx.needs_input_grad
        d1 = torch.empty_like(out1, memory_format=torch.contiguous_format) if need[0] else None
        d2 = torch.empty_like(out2, memory_format=torch.contiguous_format) if need[1] else None
        dt = torch.empty_like(tar, memory_format=torch.contiguous_format) if need[2] else None
        check(lib().se_labelembed_loss_bwd(ptr(out1), out1.stride(0), ptr(out2), out2.stride(0), ptr(tar), tar.stride(0),
                                           ptr(targets), ptr(grad), 0.0, B, C, tau, alpha, beta, ptr(aux),
                                           ptr(d1), C, ptr(d2), C, ptr(dt), C, stream_ptr()), "se_labelembed_loss_bwd")
        return d1, d2, dt, None, None, None, None


def labelembed_loss(out1, out2, tar, targets, tau=2.0, alpha=0.9, beta=0.5):
    """Per-sample label-embedding loss [B] (learn_labelembedding.py:21-37), differentiable w.r.t. out1, out2, tar."""
    return _LabelEmbedLoss.apply(out1, out2, tar, targets, tau, alpha, beta)


# --------------------------------------------------------------------------------------------
# retrieval side
class _DeviseLoss(torch.autograd.Function):
    """reference: utils.devise_ranking_loss (utils.py:103-122) and its TF-autodiff backward w.r.t. y_pred."""

    @staticmethod
    def forward(ctx, y_pred, target, embedding, margin):
        require_gpu(y_pred, target, embedding)
        yp = _rows(y_pred.to(torch.float32), "y_pred")
        yp = yp if yp.stride(1) == 1 else yp.contiguous()
        _f32_rows(embedding, "embedding")
        B, D = yp.shape
        C = embedding.shape[0]
        if embedding.shape[1] != D or embedding.device != yp.device:
            raise SehipError("embedding must be a float32 [C, %d] tensor on %s" % (D, yp.device))
        if target.dim() == 1 and not target.is_floating_point():
            labels, yt, ldt = target.long().contiguous(), None, 0
            if labels.numel() != B:
                raise SehipError("labels must be an integer [B] tensor")
        else:
            labels, yt = None, _rows(target.to(torch.float32).contiguous(), "y_true")
            ldt = yt.stride(0)
        loss_i = torch.empty((B,), dtype=torch.float32, device=yp.device)
        aux = torch.empty((max(int(lib().se_devise_aux_floats(B, C)), 1),), dtype=torch.float32, device=yp.device)
        check(lib().se_devise_loss_fwd(ptr(yp), yp.stride(0), ptr(labels), ptr(yt), ldt, ptr(embedding), embedding.stride(0), B, D, C,
                                       ctypes.c_float(margin), ptr(loss_i), ptr(aux), stream_ptr()), "se_devise_loss_fwd")
        ctx.save_for_backward(aux, embedding, labels if labels is not None else yt)
        ctx.by_label, ctx.shape, ctx.in_dtype = labels is not None, (B, D, C), y_pred.dtype
        return loss_i

    @staticmethod
    def backward(ctx, grad_loss_i):
        aux, embedding, tgt = ctx.saved_tensors
        B, D, C = ctx.shape
        g = grad_loss_i.to(torch.float32).contiguous()
        dp = torch.empty((B, D), dtype=torch.float32, device=g.device)
        labels, yt = (tgt, None) if ctx.by_label else (None, tgt)
        check(lib().se_devise_loss_bwd(ptr(labels), ptr(yt), 0 if yt is None else yt.stride(0), ptr(embedding), embedding.stride(0),
                                       ptr(g), ctypes.c_float(1.0), B, D, C, ptr(aux), ptr(dp), D, stream_ptr()), "se_devise_loss_bwd")
        return dp.to(ctx.in_dtype), None, None, None


DEVISE_TORCH_ABOVE = 1 << 29      # B * C * D from which the loss is written with PyTorch ops instead of the fused kernels


def devise_ranking_loss(y_pred, target, embedding, margin=0.1):
    """Per-sample DeViSE ranking loss [B] (utils.py:103-122), differentiable w.r.t. ``y_pred``; ``target`` = int64 labels [B]
    (rows of ``embedding`` gathered on the device) or an explicit float ``y_true`` [B, D].

    Fused fp32-MFMA kernels (``se_devise_loss_fwd/bwd``) up to ``B * C * D < DEVISE_TORCH_ABOVE``; larger problems -- batch 1024 at
    C = D = 1000 -- take the same expression in PyTorch ops: there the one-wave-per-tile forward kernel (111 us) loses to
    hipBLASLt's GEMM + elementwise kernels (measured: 223 vs 206 us forward + backward), while at the training sizes (batch 128)
    the fused pair wins (165 vs 208 us at C = D = 1000; profiles/r04_d_devise_microbench.txt).  Both are float32 and
    differentiable; the kernels are what the parity tests hold to the reference-produced fixtures."""
    B, D = y_pred.shape
    if B * embedding.shape[0] * D >= DEVISE_TORCH_ABOVE and y_pred.is_cuda:
        yp = y_pred.to(torch.float32)
        yt = embedding[target.long().clamp(0, embedding.shape[0] - 1)] if (target.dim() == 1 and not target.is_floating_point()) \
            else target.to(torch.float32)
        true_sim = (yt * yp).sum(-1)
        return torch.relu(float(margin) - true_sim[:, None] + yp @ embedding.t()).sum(-1) - float(margin)
    return _DeviseLoss.apply(y_pred, target, embedding, float(margin))


# --------------------------------------------------------------------------------------------

def _f32_rows(t, what):
    if t.dtype != torch.float32:
        raise SehipError("%s must be float32" % what)
    return _rows(t, what)


def row_sqnorm(x):
    """float32 ``np.sum(x ** 2, axis=-1)``, bit-exact (evaluate_retrieval.py:61)."""
    require_gpu(x)
    _f32_rows(x, "x")
    sq = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    check(lib().se_row_sqnorm(ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(sq), stream_ptr()), "se_row_sqnorm")
    return sq


def normalize_rows_(x):
    """In-place ``x /= np.linalg.norm(x, axis=-1, keepdims=True)``, bit-exact (evaluate_retrieval.py:58)."""
    require_gpu(x)
    _f32_rows(x, "x")
    check(lib().se_normalize_rows(ptr(x), x.stride(0), x.shape[0], x.shape[1], stream_ptr()), "se_normalize_rows")
    return x


def empty_rows(q, n, dtype, device):
    """[q, n] matrix whose row pitch is a multiple of 16 bytes (a view of a wider buffer when n is not): the distance and ranking
    kernels stream rows out with 16-byte stores and fall back to element stores on unaligned pitches -- 24,633 columns (odd): distances
    1.50 -> 1.18 ms, ranking 3.37 -> 3.01 ms (tools/bench_odd_pitch.py)."""
    per16 = 16 // torch.empty((), dtype=dtype).element_size()
    pitch = (n + per16 - 1) // per16 * per16
    buf = torch.empty((q, pitch), dtype=dtype, device=device)
    return buf if pitch == n else buf[:, :n]


def pairwise_dist(a, b=None, metric=METRIC_COSINE, sqa=None, sqb=None, kblocks=None, out=None):
    """All-pairs distances [q, n] (evaluate_retrieval.py:59 / :61-62) with the canonical FMA chain."""
    b = a if b is None else b
    require_gpu(a, b, sqa, sqb, out)
    _f32_rows(a, "a"); _f32_rows(b, "b")
    q, d = a.shape
    n = b.shape[0]
    if b.shape[1] != d:
        raise SehipError("a and b must have the same number of columns")
    if metric == METRIC_EUCLID:
        if sqa is None:
            sqa = row_sqnorm(a)
        if sqb is None:
            sqb = sqa if b is a else row_sqnorm(b)
    if out is None:
        out = empty_rows(q, n, torch.float32, a.device)
    kb, nkb = _kblocks_arg(kblocks)
    check(lib().se_pairwise_dist(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(sqa), ptr(sqb), q, n, d, int(metric),
                                 kb, nkb, ptr(out), out.stride(0), stream_ptr()), "se_pairwise_dist")
    return out


_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only per-device scratch buffer (the C ABI never allocates)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _ws_cache.pop(key, None)
        ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


_rank_ready = set()


def rank_rows_init(device=None):
    """``se_rank_rows_init`` on ``device`` (default: current): capability probe + self-test of every hardware-ordered ranking kernel
    variant; the one synchronising call of the ranking.  ``rank_rows`` calls it by itself before its first ranking on a device, so
    that every later ``se_rank_rows`` is purely asynchronous (graph-capturable)."""
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(key):
        ws = torch.empty((int(lib().se_rank_rows_init_workspace_bytes()),), dtype=torch.uint8, device=dev)
        check(lib().se_rank_rows_init(ptr(ws), ws.numel(), stream_ptr()), "se_rank_rows_init")
    _rank_ready.add(key)


def workspace_bytes(device=None):
    """Bytes the per-device workspace cache currently holds."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ws = _ws_cache.get(dev.index if dev.index is not None else torch.cuda.current_device())
    return 0 if ws is None else int(ws.numel())


def release_workspace(device=None):
    """Drop the grow-only scratch buffer of ``device`` (default: every device): the long-row ranking grows it to ~3 GB, the fused
    top-k to ~4 GB, and it is otherwise kept for the life of the process."""
    if device is None:
        _ws_cache.clear()
    else:
        dev = torch.device(device)
        _ws_cache.pop(dev.index if dev.index is not None else torch.cuda.current_device(), None)


def phase_timing(on=True):
    """``se_phase_timing``: switch the library's phase events on / off (a measuring aid: bench.py's per-leg rooflines)."""
    check(lib().se_phase_timing(1 if on else 0), "se_phase_timing")


def phase_timing_read():
    """``se_phase_timing_read`` -> (dict phase -> total ms since the last read, counters or None).  counters = {"redone", "recomputed",
    "candidates", "queries"} of the last ``retrieve_topk`` call (its workspace -- the per-device cache -- is still alive)."""
    cap = 96
    names = (ctypes.c_char_p * cap)()
    ms = (ctypes.c_float * cap)()
    cnt = (ctypes.c_int64 * 5)()
    n = lib().se_phase_timing_read(names, ms, cap, cnt)
    if n < 0:
        check(n, "se_phase_timing_read")
    out = {}
    for i in range(n):
        out[names[i].decode()] = out.get(names[i].decode(), 0.0) + float(ms[i])
    counters = None if cnt[4] < 0 else {"redone": int(cnt[1]), "recomputed": int(cnt[2]), "candidates": int(cnt[3]), "queries": int(cnt[4])}
    return out, counters


def rank_rows_workspace_bytes(q, n):
    return int(lib().se_rank_rows_workspace_bytes(int(q), int(n)))


RANK_U16_MAX_N = 53248     # rows the register-resident ranking kernel takes: the only ones it writes 16-bit ranks for


def _rank_width_code(t):
    """Index width code of se_rank_rows / se_rank_rows_check for a rank tensor: int32 -> 0, int64 -> 1, int16 (the BIT PATTERN of
    uint16 gallery indices: torch has no full uint16) -> 2."""
    if t.dtype == torch.int32:
        return 0
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int16:
        return 2
    raise SehipError("ranks must be int32, int64 or int16 (uint16 bit patterns), not %s" % t.dtype)


def rank_rows(pdist, idx64=False, out=None, idx16=False):
    """Canonical ``np.argsort(pdist, axis=-1)`` (evaluate_retrieval.py:67): (distance, index) ascending.
    ``idx16``: uint16 ranks (returned as an int16 tensor holding their bit patterns; rows of at most 53,248 columns) -- half the
    bytes for ``hierarchical_precision`` to read."""
    require_gpu(pdist, out)
    _f32_rows(pdist, "pdist")
    if (pdist.device.index if pdist.device.index is not None else torch.cuda.current_device()) not in _rank_ready:
        rank_rows_init(pdist.device)
    q, n = pdist.shape
    if out is None:
        out = empty_rows(q, n, torch.int16 if idx16 else (torch.int64 if idx64 else torch.int32), pdist.device)
    need = lib().se_rank_rows_workspace_bytes(q, n)
    ws = _workspace(need, pdist.device)
    check(lib().se_rank_rows(ptr(pdist), pdist.stride(0), q, n, ptr(out), _rank_width_code(out),
                             out.stride(0), ptr(ws), ws.numel(), stream_ptr()), "se_rank_rows")
    return out


def rank_rows_check(pdist, rank):
    """Order guard (``se_rank_rows_check``): number of rows of ``rank`` that are not the canonical ranking of ``pdist`` as far as
    adjacent entries can tell -- 0 for every output of ``rank_rows``.  Synchronises the stream."""
    require_gpu(pdist, rank)
    _f32_rows(pdist, "pdist")
    if rank.dtype not in (torch.int32, torch.int64, torch.int16) or rank.stride(1) != 1 or rank.shape != pdist.shape:
        raise SehipError("rank must be an int32 / int64 / int16 (uint16 bit patterns) matrix of pdist's shape with contiguous rows")
    q, n = pdist.shape
    ws = torch.empty((int(lib().se_rank_rows_check_workspace_bytes()),), dtype=torch.uint8, device=pdist.device)
    bad = ctypes.c_int64(0)
    check(lib().se_rank_rows_check(ptr(pdist), pdist.stride(0), q, n, ptr(rank), _rank_width_code(rank), rank.stride(0),
                                   ptr(ws), ws.numel(), ctypes.byref(bad), stream_ptr()), "se_rank_rows_check")
    return int(bad.value)


def topk_rows(pdist, k, col_offset=0):
    """k nearest columns per row under the canonical order -> (dist [q,k] f32, idx [q,k] i32)."""
    require_gpu(pdist)
    _f32_rows(pdist, "pdist")
    q, n = pdist.shape
    od = torch.empty((q, k), dtype=torch.float32, device=pdist.device)
    oi = torch.empty((q, k), dtype=torch.int32, device=pdist.device)
    check(lib().se_topk_rows(ptr(pdist), pdist.stride(0), q, n, int(col_offset), int(k), ptr(od), ptr(oi),
                             stream_ptr()), "se_topk_rows")
    return od, oi


def topk_merge(d, idx=None):
    """Merge per-shard lists [parts, q, k] (e.g. an all-gather result) into the global top-k.  With ``idx=None``, ``d`` is a PACKED
    int32 buffer [parts, 2, q, k] -- per part the float32 distance bits followed by the indices, the receive buffer of ONE
    all-gather (``se_topk_merge_packed``)."""
    require_gpu(d, idx)
    if idx is None:
        if d.dim() != 4 or d.shape[1] != 2 or d.dtype != torch.int32:
            raise SehipError("packed lists must be an int32 tensor [parts, 2, q, k]")
        d = d.contiguous()
        parts, _, q, k = d.shape
        od = torch.empty((q, k), dtype=torch.float32, device=d.device)
        oi = torch.empty((q, k), dtype=torch.int32, device=d.device)
        check(lib().se_topk_merge_packed(ptr(d), parts, q, k, ptr(od), ptr(oi), stream_ptr()), "se_topk_merge_packed")
        return od, oi
    d = d.contiguous(); idx = idx.contiguous()
    parts, q, k = d.shape
    od = torch.empty((q, k), dtype=torch.float32, device=d.device)
    oi = torch.empty((q, k), dtype=torch.int32, device=d.device)
    check(lib().se_topk_merge(ptr(d), ptr(idx), parts, q, k, ptr(od), ptr(oi), stream_ptr()), "se_topk_merge")
    return od, oi


def _kblocks_arg(kblocks):
    if kblocks is None or len(kblocks) <= 1:
        return None, 0
    return (ctypes.c_int32 * len(kblocks))(*[int(v) for v in kblocks]), len(kblocks)


def retrieve_topk(queries, gallery, k, metric=METRIC_COSINE, col_offset=0, sqq=None, sqg=None, kblocks=None, out=None):
    """Fused distances + top-k (``se_retrieve_topk``): the first k entries of every query's canonical ranking against ``gallery``
    without the [q, n] matrix; ``kblocks`` = the BLAS K-block list of ``pairwise_dist`` (D > 448).  ``out``: optional
    ``(dist f32 [q, k], idx i32 [q, k])`` contiguous destination tensors (e.g. the two halves of a packed all-gather send buffer)."""
    require_gpu(queries, gallery, sqq, sqg)
    _f32_rows(queries, "queries"); _f32_rows(gallery, "gallery")
    q, d = queries.shape
    n = gallery.shape[0]
    if gallery.shape[1] != d:
        raise SehipError("queries and gallery must have the same number of columns")
    if metric == METRIC_EUCLID:
        sqq = row_sqnorm(queries) if sqq is None else sqq
        sqg = row_sqnorm(gallery) if sqg is None else sqg
    if out is None:
        od = torch.empty((q, k), dtype=torch.float32, device=queries.device)
        oi = torch.empty((q, k), dtype=torch.int32, device=queries.device)
    else:
        od, oi = out
        require_gpu(od, oi)
        if od.dtype != torch.float32 or oi.dtype != torch.int32 or tuple(od.shape) != (q, k) or tuple(oi.shape) != (q, k) \
                or not od.is_contiguous() or not oi.is_contiguous():
            raise SehipError("out must be contiguous (float32 [q, k], int32 [q, k]) tensors")
    need = lib().se_retrieve_topk_workspace_bytes(q, n, d, gallery.stride(0), int(k))
    ws = _workspace(need, queries.device)
    kb, nkb = _kblocks_arg(kblocks)
    check(lib().se_retrieve_topk(ptr(queries), queries.stride(0), ptr(gallery), gallery.stride(0), ptr(sqq), ptr(sqg),
                                 q, n, d, int(metric), kb, nkb, int(col_offset), int(k), ptr(od), ptr(oi), ptr(ws), ws.numel(),
                                 stream_ptr()), "se_retrieve_topk")
    return od, oi


def _check_curves(*tables):
    for t in tables:
        if t.dtype != torch.float64 or t.dim() != 2 or t.stride(1) != 1:
            raise SehipError("the similarity tables / best curves must be float64 matrices with contiguous rows")


class HprecCurves:
    """Result of ``hprec_reciprocal_curves``: the table and the list length it covers."""
    __slots__ = ("data", "list_len")

    def __init__(self, data, list_len):
        self.data, self.list_len = data, int(list_len)


def hprec_reciprocal_curves(best_wup, best_lcs, list_len=None):
    """The best-possible curves pre-divided and laid out for ``hierarchical_precision`` (``se_hprec_reciprocal_curves``): once per
    gallery.  best_* [C, >=L] f64 -> ``HprecCurves`` (f64 [C, 2, curve_len(L), 2]); pass it as ``curves=``."""
    require_gpu(best_wup, best_lcs)
    _check_curves(best_wup, best_lcs)
    if best_wup.shape != best_lcs.shape or best_wup.stride(0) != best_lcs.stride(0):
        raise SehipError("best_wup and best_lcs must have the same shape and row stride")
    L = best_wup.shape[1] if list_len is None else int(list_len)
    C = best_wup.shape[0]
    out = torch.empty((C, 2, int(lib().se_hprec_curve_len(L)), 2), dtype=torch.float64, device=best_wup.device)
    check(lib().se_hprec_reciprocal_curves(ptr(best_wup), ptr(best_lcs), best_wup.stride(0), C, L, ptr(out), stream_ptr()),
          "se_hprec_reciprocal_curves")
    return HprecCurves(out, L)


def hierarchical_precision(rank, cls, qcls, qidx, wup, lcs, best_wup, best_lcs, ks, ahp_len=-1, want_ap=False, list_len=None, curves=None,
                           class_order=None):
    """Per-query hierarchical precision metrics from device rankings (class_hierarchy.py:211-316).

    rank [Q, >=L] int32, cls [N] int32, qcls [Q] int32, qidx [Q] int32 | None, wup / lcs [C, C] f64,
    best_* [C, >=L] f64, ks [nk] int32; ``curves`` = ``hprec_reciprocal_curves(best_wup, best_lcs)`` (built here when omitted: callers
    that evaluate tile after tile build it once).  ``class_order``: visit the queries class by class (keeps the best curve in L2; same results) -- by default for
    lists of 4096 ranks and more, where it pays for the counting sort.
    Returns f64 [Q, 2 nk + 3]: P@k (WUP), P@k (LCS_HEIGHT), AHP (WUP), AHP (LCS_HEIGHT), AP."""
    require_gpu(rank, cls, qcls, wup, lcs, best_wup, best_lcs, ks)
    if rank.dtype not in (torch.int32, torch.int16) or rank.stride(1) != 1:
        raise SehipError("rank must be int32 (or int16: the uint16 bit patterns rank_rows(idx16=True) writes) with contiguous rows")
    for t, name in ((cls, "cls"), (qcls, "qcls"), (ks, "ks")):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise SehipError("%s must be contiguous int32" % name)
    _check_curves(wup, lcs, best_wup, best_lcs)
    Q = rank.shape[0]
    L = rank.shape[1] if list_len is None else int(list_len)
    C = wup.shape[0]
    nk = ks.numel()
    if curves is None:
        curves = hprec_reciprocal_curves(best_wup, best_lcs, min(L, best_wup.shape[1]))
    if not isinstance(curves, HprecCurves) or curves.data.shape[0] != best_wup.shape[0] or curves.data.device != rank.device:
        raise SehipError("curves must come from hprec_reciprocal_curves for these best curves")
    out = torch.zeros((Q, 2 * nk + 3), dtype=torch.float64, device=rank.device)
    if class_order is None:
        class_order = L >= 4096
    order_ws = torch.empty((int(lib().se_hprec_order_workspace_bytes(Q)),), dtype=torch.uint8, device=rank.device) if class_order else None
    entry = lib().se_hierarchical_precision_r16 if rank.dtype == torch.int16 else lib().se_hierarchical_precision
    check(entry(ptr(rank), rank.stride(0), Q, L, ptr(cls), cls.numel(), ptr(qcls), ptr(qidx), ptr(wup), ptr(lcs), C,
                ptr(curves.data), curves.list_len, ptr(ks), nk,
                int(ahp_len), int(bool(want_ap)), ptr(out), out.stride(0), ptr(order_ws), stream_ptr()),
          "se_hierarchical_precision")
    return out
