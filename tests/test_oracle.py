"""CPU: the oracle against the golden vectors produced by the imported reference, and the two
restatements (explicit-rounding C vs literal NumPy) against each other."""
import glob
import os

import numpy as np
import pytest

from oracle import loss_oracle as lo
from oracle import retrieval_oracle as ro

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "retrieval_*.npz")))


def test_golden_fixtures_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("path", CASES)
def test_canonical_oracle_reproduces_reference_rankings(path):
    g = np.load(path)
    feats, norm = g["features"], bool(g["normalize"])
    ref = g["ref_ranking"].astype(np.int64)
    pd, rk = ro.canon_retrieval(feats, norm)
    rk = rk.astype(np.int64)
    if "ids" in g.files:
        pos = {int(v): i for i, v in enumerate(g["ids"])}
        ref = np.vectorize(pos.get)(ref)
    diff_rows = np.nonzero((rk != ref).any(axis=1))[0]
    for r in diff_rows:      # only permutations inside exact-tie groups (np.argsort is unstable)
        assert np.array_equal(pd[r][rk[r]], pd[r][ref[r]])
    if not norm or "cluster" in path:
        return
    assert len(diff_rows) == 0   # Gaussian cosine cases have no exact ties -> identical


@pytest.mark.parametrize("path", CASES)
def test_numpy_restatement_equals_c_restatement(path):
    g = np.load(path)
    feats, norm = g["features"], bool(g["normalize"])
    pd_c, rk_c = ro.canon_retrieval(feats, norm)
    if not ro.probe_host_blas_is_fma_chain(feats.shape[1]):
        pytest.skip("host BLAS does not use a sequential FMA chain for this depth")
    assert np.array_equal(ro.pdist_numpy(feats.copy(), norm), pd_c)
    assert np.array_equal(ro.pairwise_retrieval_numpy(feats.copy(), norm), rk_c)


@pytest.mark.parametrize("d", [1, 7, 8, 9, 100, 128, 129, 555, 1000, 4097])
def test_pairwise_row_sums_match_numpy(d):
    x = np.random.default_rng(d).standard_normal((23, d)).astype(np.float32)
    assert np.array_equal(ro.canon_row_sqsum(x), np.sum(x ** 2, axis=-1))
    y = x.copy()
    y /= np.linalg.norm(y, axis=-1, keepdims=True)
    assert np.array_equal(ro.canon_normalize_rows(x), y)


def test_canonical_order_rules():
    pd = np.array([[0.0, -0.0, np.nan, 1.0, -1.0, np.inf, -np.inf, 1.0, np.nan]], dtype=np.float32)
    assert ro.canon_rank_rows(pd)[0].tolist() == [6, 4, 0, 1, 3, 7, 5, 2, 8]
    d, i = ro.canon_topk_rows(pd, 4, col_offset=10)
    assert i[0].tolist() == [16, 14, 10, 11]


def test_topk_merge_is_shard_invariant():
    rng = np.random.default_rng(0)
    pd = rng.integers(0, 5, size=(6, 40)).astype(np.float32)
    want = ro.canon_topk_rows(pd, 7)
    for parts in (2, 4, 5):
        b = np.linspace(0, 40, parts + 1).astype(int)
        ds, is_ = zip(*[ro.canon_topk_rows(pd[:, b[p]:b[p + 1]], 7, col_offset=b[p]) for p in range(parts)])
        got = ro.canon_topk_merge(np.stack(ds), np.stack(is_))
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])


def test_kblocks_change_rounding_but_not_much():
    x = np.random.default_rng(1).standard_normal((40, 1000)).astype(np.float32)
    one = ro.canon_pdist(x, None, ro.METRIC_DOT)
    blk = ro.canon_pdist(x, None, ro.METRIC_DOT, kblocks=[448, 276, 276])
    assert not np.array_equal(one, blk)
    assert np.allclose(one, blk, rtol=1e-5, atol=1e-4)


# ---------------------------------------------------------------- loss oracle (unpinned: cross-check vs torch autograd)

def test_loss_oracle_against_torch_autograd():
    import torch
    rng = np.random.default_rng(0)
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    x = rng.standard_normal((32, 100))
    y = rng.integers(0, 100, size=32)
    w = rng.standard_normal(32)
    xt = torch.tensor(x, requires_grad=True)
    xh = xt * torch.rsqrt(torch.clamp((xt * xt).sum(-1, keepdim=True), min=1e-12))
    li = 1 - (torch.tensor(E)[y] * xh).sum(-1)
    (li * torch.tensor(w)).sum().backward()
    fwd = lo.cosine_loss_fwd(x, y, E)
    assert np.allclose(fwd["loss_i"], li.detach().numpy(), atol=1e-14)
    assert np.allclose(lo.cosine_loss_bwd(x, y, E, w), xt.grad.numpy(), atol=1e-13)
    assert np.allclose(lo.l2norm(x), xh.detach().numpy(), atol=1e-15)
    assert np.allclose(lo.inv_correlation(E[y], lo.l2norm(x)), fwd["loss_i"])


def test_loss_golden_is_stable():
    g = np.load(os.path.join(GOLDEN, "loss_cifar100.npz"))
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    fwd = lo.cosine_loss_fwd(g["x"], g["labels"], E)
    assert np.array_equal(fwd["loss_i"], g["loss_i"])
    assert np.array_equal(lo.cosine_loss_bwd(g["x"], g["labels"], E, np.full(128, 1 / 128)), g["dx"])
    acc = lo.nn_accuracy(E, True)(E[g["labels"]], fwd["xhat"])
    assert np.array_equal(acc, g["acc"])


def test_nn_accuracy_topk_and_euclid_semantics():
    E = np.eye(4)
    p = np.array([[0.9, 0.1, 0, 0], [0.1, 0.9, 0, 0], [0.3, 0.5, 0.6, 0]])
    y = np.array([0, 0, 1])
    assert lo.nn_accuracy(E, True)(E[y], p).tolist() == [1, 0, 0]
    assert lo.nn_accuracy(E, True, k=2)(E[y], p).tolist() == [1, 1, 1]
    assert lo.nn_accuracy(E, False)(E[y], p).tolist() == [1, 0, 0]
    assert lo.nn_accuracy(E, False, k=2)(E[y], p).tolist() == [1, 1, 1]


def test_labelembed_loss_matches_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    B, C = 16, 10
    o1, o2, tar = (rng.standard_normal((B, C)) for _ in range(3))
    t = rng.integers(0, C, size=B)
    got = lo.labelembed_loss(o1, o2, tar, t)
    O1, O2, T = (torch.tensor(a) for a in (o1, o2, tar))
    tt = torch.tensor(t)
    p2 = F.softmax(O2, -1)
    mask = (O2.argmax(-1) == tt).double()
    ref = (0.5 * F.cross_entropy(O1, tt, reduction="none") + 0.5 * -(F.softmax(T, -1) * F.log_softmax(O1, -1)).sum(1)
           + F.cross_entropy(O2, tt, reduction="none")
           + -(F.softmax(O2 / 2, -1) * F.log_softmax(T, -1)).sum(1) * mask * (B / (mask.sum() + 1e-8))
           + torch.relu(p2[torch.arange(B), tt] - 0.9))
    assert np.allclose(got, ref.numpy(), atol=1e-6)


def test_labelembed_oracle_backward_matches_torch_autograd():
    """The closed-form label-embedding gradient of the oracle == autograd of the reference expression
    (learn_labelembedding.py:21-37 restated in torch with detach() for stop_gradient)."""
    import torch
    from oracle import loss_oracle as lo
    rng = np.random.default_rng(5)
    b, c = 16, 37
    o1, o2, tr = (rng.standard_normal((b, c)) * 2 for _ in range(3))
    y = rng.integers(0, c, size=b)
    o2[np.arange(b)[::2], y[::2]] += 6.0           # half of the samples are classified correctly by out2 (mask = 1)
    g = rng.standard_normal(b)
    t1, t2, t3 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (o1, o2, tr))
    ty = torch.tensor(y)
    tau, alpha, beta = 2.0, 0.9, 0.5
    out2_prob = torch.softmax(t2, dim=1)
    tau2_prob = torch.softmax(t2 / tau, dim=1).detach()
    soft_tar = torch.softmax(t3, dim=1).detach()
    rows = torch.arange(b)
    l_o1_y = -torch.log(torch.clamp(torch.softmax(t1, dim=1), 1e-7, 1 - 1e-7)[rows, ty])
    mask = (t2.argmax(dim=1) == ty).double().detach()
    l_o1_emb = -torch.sum(soft_tar * torch.log_softmax(t1, dim=1), dim=1)
    l_o2_y = -torch.log(torch.clamp(out2_prob, 1e-7, 1 - 1e-7)[rows, ty])
    l_emb_o2 = -torch.sum(tau2_prob * torch.log_softmax(t3, dim=1), dim=1) * mask * (b / (mask.sum() + 1e-8))
    l_re = torch.relu(out2_prob[rows, ty] - alpha)
    loss = beta * l_o1_y + (1 - beta) * l_o1_emb + l_o2_y + l_emb_o2 + l_re
    assert np.allclose(loss.detach().numpy(), lo.labelembed_loss(o1, o2, tr, y, tau, alpha, beta), atol=1e-12)
    loss.backward(torch.tensor(g))
    d1, d2, dt = lo.labelembed_loss_bwd(o1, o2, tr, y, g, tau, alpha, beta)
    assert np.allclose(t1.grad.numpy(), d1, atol=1e-12)
    assert np.allclose(t2.grad.numpy(), d2, atol=1e-12)
    assert np.allclose(t3.grad.numpy(), dt, atol=1e-12)
