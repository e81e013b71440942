"""oracle/loss_oracle.py -- TEST INFRASTRUCTURE ONLY.

NumPy restatement (float64 by default, any float dtype on request) of the reference's
training-side hot path: L2-normalisation head, cosine loss, its closed-form backward, and the
nearest-class-embedding accuracy metrics.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Parity status: PINNED to the reference's own source lines.  tests/golden/loss_ref_*.npz hold the
outputs of the reference's utils.py:34-127 and learn_labelembedding.py:17-37, imported UNMODIFIED
(oracle/ref_import.py) and evaluated on a NumPy ``keras.backend`` stand-in (oracle/keras_stub.py) in
float32 and float64; tests/test_oracle.py checks every function below against them (float64: to
1e-12).  What remains a restatement of third-party code is listed in oracle/keras_stub.py:
``tf.nn.l2_normalize`` (``x * rsqrt(maximum(sum(square(x)), 1e-12))``), ``tf.nn.top_k``,
``tf.nn.log_softmax`` / ``K.softmax`` and Keras 2.2's ``sparse_categorical_crossentropy`` -- Keras /
TensorFlow themselves (README.md:315-316 of the reference) are not installable here.  The backward
formulas are closed forms checked against torch-float64 autograd of the same expressions.

Citations are into /root/reference/.
"""
import numpy as np


# ---------------------------------------------------------------------------- head + losses

def l2norm(x, eps=1e-12):
    """utils.py:125-127 -> tf.nn.l2_normalize(x, -1): x * rsqrt(max(sum(x^2), eps))."""
    x = np.asarray(x)
    ss = np.sum(np.square(x), axis=-1, keepdims=True)
    return x * (1.0 / np.sqrt(np.maximum(ss, eps)))


def inv_correlation(y_true, y_pred):
    """utils.py:44-46: 1 - sum(y_true * y_pred, -1)."""
    return 1.0 - np.sum(np.asarray(y_true) * np.asarray(y_pred), axis=-1)


def squared_distance(y_true, y_pred):
    """utils.py:34-36."""
    return np.sum(np.square(np.asarray(y_pred) - np.asarray(y_true)), axis=-1)


def mean_distance(y_true, y_pred):
    """utils.py:39-41."""
    return np.sqrt(squared_distance(y_true, y_pred))


def transform_inputs(labels, embedding):
    """learn_image_embeddings.py:48-50: y_true = embedding[y] (host gather)."""
    return np.asarray(embedding)[np.asarray(labels)]


def cosine_loss_fwd(x, labels, embedding, dtype=np.float64, eps=1e-12):
    """l2norm head + inv_correlation against gathered class embeddings + Keras batch mean.

    Follows learn_image_embeddings.py:127-128 (Lambda(l2norm)), :48-50 (gather), utils.py:44-46.
    Returns dict(xhat [B,D], inv_norm [B], loss_i [B], loss scalar)."""
    x = np.asarray(x, dtype=dtype)
    e = np.asarray(embedding, dtype=dtype)
    t = e[np.asarray(labels)]
    ss = np.sum(x * x, axis=-1)
    inv_norm = 1.0 / np.sqrt(np.maximum(ss, dtype(eps)))
    xhat = x * inv_norm[:, None]
    loss_i = 1.0 - np.sum(t * xhat, axis=-1)
    return {"xhat": xhat, "inv_norm": inv_norm, "loss_i": loss_i, "loss": loss_i.mean()}


def cosine_loss_bwd(x, labels, embedding, grad_loss_i, dtype=np.float64, eps=1e-12):
    """Closed-form d(sum_i grad_loss_i * loss_i)/dx (what TF autodiff produces for the ops above).

    g_i = -grad_loss_i * E[y_i];  where sum(x^2) >= eps:  dx = (g - xhat * (xhat . g)) * inv_norm,
    where sum(x^2) < eps (the max() clamps, so inv_norm is a constant 1/sqrt(eps)):  dx = g * inv_norm."""
    x = np.asarray(x, dtype=dtype)
    e = np.asarray(embedding, dtype=dtype)
    t = e[np.asarray(labels)]
    w = np.asarray(grad_loss_i, dtype=dtype)
    ss = np.sum(x * x, axis=-1)
    inv_norm = 1.0 / np.sqrt(np.maximum(ss, dtype(eps)))
    xhat = x * inv_norm[:, None]
    g = -w[:, None] * t
    proj = np.sum(xhat * g, axis=-1, keepdims=True)
    dx_reg = (g - xhat * proj) * inv_norm[:, None]
    dx_clamped = g * inv_norm[:, None]
    return np.where((ss >= eps)[:, None], dx_reg, dx_clamped)


# ---------------------------------------------------------------------------- metrics

def nn_accuracy(embedding, dot_prod_sim=False, k=1, dtype=np.float64):
    """utils.py:57-100.  Returns f(y_true, y_pred) -> [B] of 0/1 like the Keras metric."""
    e = np.asarray(embedding, dtype=dtype)

    def euclid_acc(y_true, y_pred):                       # utils.py:73-85
        y_true = np.asarray(y_true, dtype=dtype)
        y_pred = np.asarray(y_pred, dtype=dtype)
        centroids = e.T
        centroids_norm = (centroids ** 2).sum(axis=0, keepdims=True)
        pred_norm = np.sum(np.square(y_pred), axis=1, keepdims=True)
        dist = pred_norm + centroids_norm - 2 * np.dot(y_pred, centroids)
        true_dist = np.sum(np.square(y_pred - y_true), axis=-1)
        if k <= 1:
            return (np.abs(true_dist - dist.min(axis=-1)) < 1e-6).astype(dtype)
        topk = np.sort(dist, axis=-1)[:, :k]
        return np.any(np.abs(topk - true_dist[:, None]) < 1e-6, axis=-1).astype(dtype)

    def max_sim_acc(y_true, y_pred):                      # utils.py:87-95
        y_true = np.asarray(y_true, dtype=dtype)
        y_pred = np.asarray(y_pred, dtype=dtype)
        sim = np.dot(y_pred, e.T)
        true_sim = np.sum(y_pred * y_true, axis=-1)
        if k <= 1:
            return (np.abs(sim.max(axis=-1) - true_sim) < 1e-6).astype(dtype)
        topk = -np.sort(-sim, axis=-1)[:, :k]
        return np.any(np.abs(topk - true_sim[:, None]) < 1e-6, axis=-1).astype(dtype)

    return max_sim_acc if dot_prod_sim else euclid_acc


def class_scores(y_pred, embedding, dot_prod_sim=True, dtype=np.float64):
    """The dense contraction inside the metric: sim = y_pred @ E^T (utils.py:90) or the squared
    distances of utils.py:75-78."""
    p = np.asarray(y_pred, dtype=dtype)
    e = np.asarray(embedding, dtype=dtype)
    if dot_prod_sim:
        return p @ e.T
    return np.sum(p * p, axis=1, keepdims=True) + np.sum(e * e, axis=1)[None, :] - 2 * (p @ e.T)


def devise_ranking_loss(embedding, margin=0.1, dtype=np.float64):
    """utils.py:103-122."""
    e = np.asarray(embedding, dtype=dtype)

    def _loss(y_true, y_pred):
        y_true = np.asarray(y_true, dtype=dtype)
        y_pred = np.asarray(y_pred, dtype=dtype)
        true_sim = np.sum(y_true * y_pred, axis=-1)
        other_sim = y_pred @ e.T
        return np.sum(np.maximum(margin - true_sim[:, None] + other_sim, 0), axis=-1) - margin

    return _loss


# ---------------------------------------------------------------------------- label-embedding loss

def _softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    ez = np.exp(z)
    return ez / ez.sum(axis=-1, keepdims=True)


def _log_softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def _keras_sparse_ce(prob, targets, keps=1e-7):
    """Keras 2.2 ``K.sparse_categorical_crossentropy(target, output)`` on probabilities: clip to [1e-7, 1 - 1e-7], log,
    then TF's sparse_softmax_cross_entropy_with_logits -- which renormalises: -(log c_y - log sum_j c_j)."""
    c = np.clip(prob, keps, 1 - keps)
    return -(np.log(c[np.arange(len(targets)), targets]) - np.log(c.sum(axis=-1)))


def labelembed_loss(out1, out2, tar, targets, tau=2.0, alpha=0.9, beta=0.5, dtype=np.float64):
    """learn_labelembedding.py:17-37 (forward value only; stop_gradient has no forward effect)."""
    out1 = np.asarray(out1, dtype=dtype)
    out2 = np.asarray(out2, dtype=dtype)
    tar = np.asarray(tar, dtype=dtype)
    targets = np.asarray(targets).astype(np.int64)
    b = out1.shape[0]
    rows = np.arange(b)

    out2_prob = _softmax(out2)
    tau2_prob = _softmax(out2 / tau)
    soft_tar = _softmax(tar)

    l_o1_y = _keras_sparse_ce(_softmax(out1), targets)
    pred = out2.argmax(axis=-1)
    mask = (pred == targets).astype(dtype)
    l_o1_emb = -np.sum(soft_tar * _log_softmax(out1), axis=1)
    l_o2_y = _keras_sparse_ce(out2_prob, targets)
    l_emb_o2 = -np.sum(tau2_prob * _log_softmax(tar), axis=1) * mask * (b / (mask.sum() + 1e-8))
    l_re = np.maximum(out2_prob[rows, targets] - alpha, 0)
    return beta * l_o1_y + (1 - beta) * l_o1_emb + l_o2_y + l_emb_o2 + l_re


def _keras_sparse_ce_grad(sm, hot, keps=1e-7):
    """d/d logits of `_keras_sparse_ce(softmax(logits), y)`: with in_j = 1{eps < p_j < 1 - eps} (clip_by_value passes the
    gradient strictly inside the range), S = sum_j clip(p_j), R = sum_j in_j p_j:
        d_i = in_y (p_i - onehot_i) + p_i (in_i - R) / S."""
    inside = ((sm > keps) & (sm < 1 - keps)).astype(sm.dtype)
    S = np.clip(sm, keps, 1 - keps).sum(axis=-1, keepdims=True)
    R = (inside * sm).sum(axis=-1, keepdims=True)
    in_y = (inside * hot).sum(axis=-1, keepdims=True)
    return in_y * (sm - hot) + sm * (inside - R) / S


def labelembed_loss_bwd(out1, out2, tar, targets, grad_loss_i, tau=2.0, alpha=0.9, beta=0.5, dtype=np.float64):
    """Closed-form gradient of `labelembed_loss` w.r.t. (out1, out2, tar) as TF autodiff derives it from
    learn_labelembedding.py:21-37: softmax(out2 / tau), softmax(tar) inside L_o1_emb and the arg-max mask are
    stop_gradient; clip_by_value passes gradient only strictly inside [1e-7, 1 - 1e-7].  Checked against torch
    autograd of the same expression in tests/test_oracle.py."""
    out1 = np.asarray(out1, dtype=dtype)
    out2 = np.asarray(out2, dtype=dtype)
    tar = np.asarray(tar, dtype=dtype)
    targets = np.asarray(targets).astype(np.int64)
    g = np.asarray(grad_loss_i, dtype=dtype)[:, None]
    b, c = out1.shape
    rows = np.arange(b)
    hot = np.zeros((b, c), dtype=dtype)
    hot[rows, targets] = 1
    sm1, sm2, smt, sm2t = _softmax(out1), _softmax(out2), _softmax(tar), _softmax(out2 / tau)
    p2y = sm2[rows, targets][:, None]
    kre = np.where(p2y > alpha, p2y, 0.0)
    mask = (out2.argmax(axis=-1) == targets).astype(dtype)
    scale = b / (mask.sum() + 1e-8)
    d1 = g * (beta * _keras_sparse_ce_grad(sm1, hot) + (1 - beta) * (sm1 - smt))
    d2 = g * (_keras_sparse_ce_grad(sm2, hot) + kre * (hot - sm2))
    dt = g * (mask * scale)[:, None] * (smt - sm2t)
    return d1, d2, dt
