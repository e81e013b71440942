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
    assert len(CASES) >= 12
    assert any("d555" in c for c in CASES) and any("d1000" in c for c in CASES)


def _kblocks(g):
    """D > 448: the K-block list of the BLAS that produced the fixture (probed by oracle/make_golden.py)."""
    return g["kblocks"].tolist() if "kblocks" in g.files else None


@pytest.mark.parametrize("path", CASES)
def test_canonical_oracle_reproduces_reference_rankings(path):
    g = np.load(path)
    feats, norm = g["features"], bool(g["normalize"])
    ref = g["ref_ranking"].astype(np.int64)
    pd, rk = ro.canon_retrieval(feats, norm, kblocks=_kblocks(g))
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


@pytest.mark.parametrize("branch", ["cos", "euc"])
def test_oracle_matches_the_larger_reference_fixture(branch):
    """tests/golden/bigretrieval_cluster.npz (round 5): 4,096 clustered items with 64 exact duplicates, D = 100 -- the rankings of 128
    query rows as the imported, unmodified reference returned them.  The canonical oracle must equal them except inside exact-tie
    groups (the reference's argsort is unstable)."""
    g = np.load(os.path.join(GOLDEN, "bigretrieval_cluster.npz"))
    feats, rows = g["features"], g["rows"]
    pd, rk = ro.canon_retrieval(feats, branch == "cos")
    ref = g["ref_ranking_rows_" + branch].astype(np.int64)
    ties = 0
    for i, r in enumerate(rows):
        mine = rk[r].astype(np.int64)
        if not np.array_equal(mine, ref[i]):
            assert np.array_equal(pd[r][mine], pd[r][ref[i]]), "row %d differs outside a tie group" % r
            ties += 1
    assert ties > 0          # the duplicate rows do force exact ties: the gate is not vacuous


@pytest.mark.parametrize("path", [c for c in CASES if "d555" in c or "d1000" in c])
def test_single_chain_does_not_reproduce_the_reference_beyond_448(path):
    """The K-block list matters: a single FMA chain over all of D ranks some near-ties differently from the reference."""
    g = np.load(path)
    pd1, _ = ro.canon_retrieval(g["features"], bool(g["normalize"]))
    pdk, _ = ro.canon_retrieval(g["features"], bool(g["normalize"]), kblocks=_kblocks(g))
    assert not np.array_equal(pd1, pdk)


@pytest.mark.parametrize("path", CASES)
def test_numpy_restatement_equals_c_restatement(path):
    g = np.load(path)
    feats, norm = g["features"], bool(g["normalize"])
    pd_c, rk_c = ro.canon_retrieval(feats, norm, kblocks=_kblocks(g))
    if _kblocks(g) is not None:     # D > 448: this host's BLAS must reproduce the fixture's K-block list (it produced it)
        x = feats[:64].copy()
        if not np.array_equal(np.dot(x, x.T), ro.canon_pdist(x, None, ro.METRIC_DOT, kblocks=_kblocks(g))):
            pytest.skip("host BLAS blocks K differently from the BLAS that produced the fixture")
    elif not ro.probe_host_blas_is_fma_chain(feats.shape[1]):
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
    o1[:, :5] -= 30.0                              # probabilities below 1e-7: the clip changes value and gradient
    o2[1::2, -5:] -= 30.0
    o2[0, y[0]] += 40.0                            # p_y above 1 - 1e-7
    g = rng.standard_normal(b)
    t1, t2, t3 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (o1, o2, tr))
    ty = torch.tensor(y)
    tau, alpha, beta = 2.0, 0.9, 0.5
    out2_prob = torch.softmax(t2, dim=1)
    tau2_prob = torch.softmax(t2 / tau, dim=1).detach()
    soft_tar = torch.softmax(t3, dim=1).detach()
    rows = torch.arange(b)
    def keras_sparse_ce(prob):   # Keras 2.2: clip, log, sparse_softmax_cross_entropy_with_logits (renormalises)
        return torch.nn.functional.cross_entropy(torch.log(torch.clamp(prob, 1e-7, 1 - 1e-7)), ty, reduction="none")
    l_o1_y = keras_sparse_ce(torch.softmax(t1, dim=1))
    mask = (t2.argmax(dim=1) == ty).double().detach()
    l_o1_emb = -torch.sum(soft_tar * torch.log_softmax(t1, dim=1), dim=1)
    l_o2_y = keras_sparse_ce(out2_prob)
    l_emb_o2 = -torch.sum(tau2_prob * torch.log_softmax(t3, dim=1), dim=1) * mask * (b / (mask.sum() + 1e-8))
    l_re = torch.relu(out2_prob[rows, ty] - alpha)
    loss = beta * l_o1_y + (1 - beta) * l_o1_emb + l_o2_y + l_emb_o2 + l_re
    assert np.allclose(loss.detach().numpy(), lo.labelembed_loss(o1, o2, tr, y, tau, alpha, beta), atol=1e-12)
    loss.backward(torch.tensor(g))
    d1, d2, dt = lo.labelembed_loss_bwd(o1, o2, tr, y, g, tau, alpha, beta)
    assert np.allclose(t1.grad.numpy(), d1, atol=1e-12)
    assert np.allclose(t2.grad.numpy(), d2, atol=1e-12)
    assert np.allclose(t3.grad.numpy(), dt, atol=1e-12)


# ---------------------------------------------------------------- loss oracle vs the reference's own source lines

LOSS_REF = sorted(glob.glob(os.path.join(GOLDEN, "loss_ref_*.npz")))


def _embedding_for(path):
    key = os.path.basename(path)[len("loss_ref_"):-len(".npz")]
    if key == "imagenet_mintree_unitsphere":
        return np.load(os.path.join(GOLDEN, "imagenet_mintree_unitsphere.npz"))["embedding"].astype(np.float64)
    return np.load(os.path.join(GOLDEN, "embeddings.npz"))[key].astype(np.float64)


def test_loss_reference_fixtures_present():
    assert len(LOSS_REF) == 5


@pytest.mark.parametrize("path", LOSS_REF)
def test_loss_oracle_equals_reference_lines(path):
    """loss_oracle == utils.py:34-127 / learn_labelembedding.py:17-37 (imported unmodified, NumPy keras backend):
    float64 evaluation to 1e-12, float32 evaluation (the reference's precision) within float32 rounding."""
    g = np.load(path)
    E = _embedding_for(path)
    x, y = g["x"], g["labels"]
    f = lo.cosine_loss_fwd(x, y, E)
    assert np.abs(f["xhat"] - g["xhat_64"]).max() <= 1e-12 * max(1.0, np.abs(g["xhat_64"]).max())
    assert np.abs(lo.l2norm(x.astype(np.float64)) - g["xhat_64"]).max() <= 1e-12 * max(1.0, np.abs(g["xhat_64"]).max())
    assert np.abs(f["loss_i"] - g["inv_correlation_64"]).max() <= 1e-12
    assert np.abs(f["loss_i"] - g["inv_correlation_32"]).max() <= 5e-6
    assert np.abs(lo.inv_correlation(E[y], g["xhat_64"]) - g["inv_correlation_64"]).max() <= 1e-12
    x64 = x.astype(np.float64)
    sq = lo.squared_distance(E[y], x64)
    assert np.abs(sq - g["squared_distance_64"]).max() <= 1e-12 * max(1.0, sq.max())
    assert np.abs(sq - g["squared_distance_32"]).max() <= 1e-5 * max(1.0, sq.max())
    assert np.abs(lo.mean_distance(E[y], x64) - g["mean_distance_64"]).max() <= 1e-12 * max(1.0, sq.max())
    dv = lo.devise_ranking_loss(E, 0.1)(E[y], f["xhat"])
    assert np.abs(dv - g["devise_ranking_loss_64"]).max() <= 1e-11 * max(1.0, np.abs(dv).max())
    assert np.abs(dv - g["devise_ranking_loss_32"]).max() <= 1e-5 * max(1.0, np.abs(dv).max())
    for k in (1, 5):
        assert np.array_equal(lo.nn_accuracy(E, True, k)(E[y], f["xhat"]), g["max_sim_acc%d_64" % k])
        assert np.array_equal(lo.nn_accuracy(E, False, k)(E[y], x64), g["nn_accuracy%d_64" % k])
    le = lo.labelembed_loss(g["le_out1"], g["le_out2"], g["le_tar"], y)
    assert np.abs(le - g["labelembed_loss_64"]).max() <= 1e-12 * max(1.0, np.abs(le).max())
    assert np.abs(le - g["labelembed_loss_32"]).max() <= 2e-5


@pytest.mark.parametrize("path", LOSS_REF)
def test_reference_metric_is_precision_sensitive_only_near_the_band(path):
    """Documents where the reference's float32 metric and its float64 evaluation disagree: only rows whose decisive
    |score - true score| sits within float32 rounding of the 1e-6 band (utils.py:84,93) -- the rows the GPU parity
    tests exclude by the same criterion."""
    g = np.load(path)
    E = _embedding_for(path)
    y = g["labels"]
    for dot, name, pred in ((True, "max_sim_acc", g["xhat_64"]), (False, "nn_accuracy", g["x"].astype(np.float64))):
        s = lo.class_scores(pred, E, dot)
        best = s.max(axis=1) if dot else s.min(axis=1)
        true = np.sum(pred * E[y], axis=1) if dot else np.sum(np.square(pred - E[y]), axis=1)
        margin = np.abs(np.abs(best - true) - 1e-6)
        noise = 4e-7 * max(1.0, np.abs(s).max()) * (1 if dot else 8)
        differ = g[name + "1_32"] != g[name + "1_64"]
        assert not np.any(differ & (margin > noise)), (name, margin[differ])


def test_lr_schedules_match_reference_trajectories():
    """utils.get_lr_schedule (host mirror) vs the reference's get_lr_schedule + clr_callback.py / sgdr_callback.py
    driven epoch by epoch (utils.py:288-399)."""
    import utils
    g = np.load(os.path.join(GOLDEN, "lr_schedules.npz"))

    class T(object):
        lr = 0.1
        is_main_process = True

    cbs, n = utils.get_lr_schedule("SGDR", 50000, 100, {"sgdr_base_len": 4, "sgdr_mul": 2, "sgdr_max_lr": 0.1})
    assert n == int(g["sgdr_epochs"])
    t = T()
    cbs[0].on_train_begin(t)
    lrs = []
    for ep in range(30):
        lrs.append(t.lr)
        cbs[0].on_epoch_end(t, ep, {})
    assert np.allclose(lrs, g["sgdr_lr_per_epoch"], rtol=1e-12, atol=0)

    cbs, n = utils.get_lr_schedule("CLR", 1000, 100, {"clr_step_len": 2, "clr_min_lr": 1e-5, "clr_max_lr": 0.1})
    assert n == int(g["clr_epochs"])
    t = T()
    cbs[0].on_train_begin(t)
    lrs = []
    for it in range(100):
        lrs.append(t.lr)
        cbs[0].on_batch_end(t, it, {})
    assert np.allclose(lrs, g["clr_lr_per_batch"], rtol=1e-12, atol=0)

    cbs, n = utils.get_lr_schedule("SGD", 50000, 100, {"sgd_schedule": "1:0.1,31:0.01,41:0.001,50"})
    assert n == int(g["sgd_schedule_epochs"])
    assert [cbs[0].schedule(ep, 0.5) for ep in range(50)] == g["sgd_schedule_lr"].tolist()
    cbs, n = utils.get_lr_schedule("ResNet-Schedule", 50000, 100, {})
    assert n == int(g["resnet_schedule_epochs"])
    assert [cbs[0].schedule(ep) for ep in range(164)] == g["resnet_schedule_lr"].tolist()
    cbs, n = utils.get_lr_schedule("SGD", 50000, 100, {})
    assert n == int(g["sgd_plateau_epochs"]) and cbs[0].patience == int(g["sgd_plateau_patience"])
    assert cbs[0].min_lr == float(g["sgd_plateau_min_lr"])


def test_imagenet_mintree_embedding_fixture():
    """The regenerated imagenet_mintree.unitsphere (compute_class_embedding.py:14-40): unit rows, lower-triangular, dot products
    = 1 - lcs_height / max_height in [0, 1]."""
    g = np.load(os.path.join(GOLDEN, "imagenet_mintree_unitsphere.npz"))
    E = g["embedding"].astype(np.float64)
    assert E.shape == (1000, 1000) and len(g["ind2label"]) == 1000 and str(g["ind2label"][0]).startswith("n")
    assert np.abs(np.linalg.norm(E, axis=1) - 1).max() < 1e-6
    assert np.abs(np.triu(E, 1)).max() == 0
    S = E @ E.T
    assert S.min() > -1e-6 and S.max() < 1 + 1e-6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "topk_head_*.npz"))))
def test_canonical_topk_reproduces_reference_heads_beyond_448(path):
    """tests/golden/topk_head_*.npz: the first 256 entries of the imported reference's rankings
    (evaluate_retrieval.py:57-67) on 640-item D = 555 / D = 1000 problems.  The oracle's top-251 with the fixture's
    K-block list equals them outside exact-tie groups; the single-chain arithmetic does not (which is why the top-k /
    sharded-gallery path has to carry the list)."""
    feat, norm, kb, head = ro.load_topk_fixture(path)
    x = ro.canon_normalize_rows(feat) if norm else feat
    metric = ro.METRIC_COSINE if norm else ro.METRIC_EUCLID
    pd = ro.canon_pdist(x, None, metric, kblocks=kb)
    d, i = ro.canon_topk_rows(pd, 251)
    assert np.array_equal(i, ro.canon_rank_rows(pd)[:, :251])
    for r in np.nonzero((i != head[:, :251]).any(axis=1))[0]:
        assert np.array_equal(pd[r][i[r]], pd[r][head[r, :251]]), "row %d differs outside a tie group" % r
    pd1 = ro.canon_pdist(x, None, metric)
    _, i1 = ro.canon_topk_rows(pd1, 251)
    assert any(not np.array_equal(pd[r][i1[r]], pd[r][head[r, :251]]) for r in range(len(x)))
