"""GPU: the drop-in Python interfaces (reference signatures) running on the HIP kernels --
evaluate_retrieval.pairwise_retrieval, utils losses/metrics, one training step of the engine."""
import glob
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo
from oracle import retrieval_oracle as ro

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "retrieval_*.npz"))))
def test_pairwise_retrieval_dropin_vs_reference_output(path):
    import evaluate_retrieval as er
    g = np.load(path)
    feats, norm = g["features"].astype(np.float32), bool(g["normalize"])
    ref = g["ref_ranking"]
    if "ids" in g.files:
        inp = {"feat": {int(i): f.copy() for i, f in zip(g["ids"], feats)}}
        ids = [int(i) for i in g["ids"]]
    else:
        inp, ids = feats.copy(), list(range(len(feats)))
    # D > 448: the fixture carries the K-block list of the BLAS that produced the reference ranking; `kblocks='openblas'`
    # (evaluate_retrieval.host_blas_kblocks) must derive the same list
    kb = g["kblocks"].tolist() if "kblocks" in g.files else None
    if kb is not None:
        assert er.host_blas_kblocks(feats.shape[1]) == kb
    got = er.pairwise_retrieval(inp, normalize=norm, return_generator=False, kblocks="openblas" if kb else None)
    assert list(got.keys()) == ids
    pd, _ = ro.canon_retrieval(feats, norm, kblocks=kb)
    pos = {v: i for i, v in enumerate(ids)}
    for r, qid in enumerate(ids):
        mine = np.array([pos[v] for v in got[qid]])
        theirs = np.array([pos[int(v)] for v in ref[r]])
        assert sorted(mine.tolist()) == list(range(len(ids)))
        assert np.array_equal(pd[r][mine], pd[r][theirs])          # equal up to order inside exact ties
    if norm and not isinstance(inp, dict):                          # the reference normalises its input in place
        assert np.array_equal(inp, ro.canon_normalize_rows(feats))


def test_pairwise_retrieval_generator_pickle_and_errors(tmp_path):
    import evaluate_retrieval as er
    rng = np.random.default_rng(0)
    feats = {i * 3: rng.standard_normal(8).astype(np.float32) for i in range(20)}
    p = tmp_path / "feat.pickle"
    with open(p, "wb") as f:
        pickle.dump({"feat": feats}, f)
    gen = er.pairwise_retrieval(str(p), normalize=True)
    first = next(gen)
    assert first[0] == 0 and first[1][0] == 0 and len(first[1]) == 20 and isinstance(first[1], list)
    with pytest.raises(ValueError):
        er.pairwise_retrieval({i: np.zeros((2, 2), np.float32) for i in range(3)})


def test_utils_losses_and_metrics_keras_signature():
    import utils
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    Ed = torch.from_numpy(E.astype(np.float32)).cuda()
    rng = np.random.default_rng(2)
    x = rng.standard_normal((64, 100)).astype(np.float32)
    y = rng.integers(0, 100, size=64)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    xn = utils.l2norm(xd)
    assert np.abs(xn.cpu().numpy() - lo.l2norm(x.astype(np.float64))).max() < 1e-6
    li = utils.inv_correlation(Ed[yd], xn)                         # reference convention: gathered y_true
    want = lo.cosine_loss_fwd(x, y, E)["loss_i"]
    assert np.abs(li.cpu().numpy() - want).max() < 1e-5
    fused = utils.CosineEmbeddingLoss(Ed)(yd, xd)
    assert np.abs(fused.cpu().numpy() - want).max() < 1e-5
    m = utils.nn_accuracy(E, dot_prod_sim=True)
    assert m.name == "max_sim_acc" and utils.nn_accuracy(E, True, 5).name == "max_sim_acc5" and utils.nn_accuracy(E).name == "nn_accuracy"
    a_lab = m(yd, xn).cpu().numpy()
    a_emb = m(Ed[yd], xn).cpu().numpy()                            # y_true as embeddings, like the reference
    assert np.array_equal(a_lab, a_emb)
    assert np.array_equal(a_lab, lo.nn_accuracy(E, True)(E[y], lo.l2norm(x.astype(np.float64))).astype(np.float32))
    # l2norm backward through autograd vs oracle closed form
    xr = xd.clone().requires_grad_(True)
    (utils.inv_correlation(Ed[yd], utils.l2norm(xr)) / 64).sum().backward()
    assert np.abs(xr.grad.cpu().numpy() - lo.cosine_loss_bwd(x, y, E, np.full(64, 1 / 64))).max() < 1e-7
    d = utils.devise_ranking_loss(E)(yd, xn).cpu().numpy()
    assert np.allclose(d, lo.devise_ranking_loss(E)(E[y], lo.l2norm(x.astype(np.float64))), atol=1e-4)
    assert np.allclose(utils.squared_distance(Ed[yd], xd).cpu().numpy(), lo.squared_distance(E[y], x), rtol=1e-5)


def test_training_step_resnet110_fc_uses_hip_loss_and_learns():
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    Ed = torch.from_numpy(E.astype(np.float32)).cuda()
    torch.manual_seed(0)
    model = utils.build_network(100, "resnet-110-fc", input_channels=3).cuda()
    loss = utils.CosineEmbeddingLoss(Ed)
    l2 = {id(p): model.regularizer for p in model.regularized_parameters()}
    tr = Trainer(model, {"l2norm": (loss, 1.0)}, {"l2norm": [utils.nn_accuracy(Ed, dot_prod_sim=True)]}, lr=0.05, clipnorm=10.0, l2_of=l2)
    gen = SyntheticGenerator(100, 32, 3, 256, 64)
    X, y = gen.train_sequence(64, shuffle=False)[0]
    logs = {}
    first = float(tr.train_step(X, y, logs).detach())
    for _ in range(14):
        last = float(tr.train_step(X, y, logs).detach())
    assert np.isfinite(last) and last < first - 0.05                # memorises the fixed batch
    assert "max_sim_acc" in logs and loss.last_normalized.shape == (64, 100)
    # the fused head agrees with the oracle on the raw outputs the backbone produced
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        raw = model(X)
    li = loss(y, raw)
    want = lo.cosine_loss_fwd(raw.float().cpu().numpy().astype(np.float64), y.cpu().numpy(), E)["loss_i"]
    assert np.abs(li.cpu().numpy() - want).max() < 1e-4
    ev = tr.evaluate(gen.test_sequence(32))
    assert set(ev) == {"loss", "max_sim_acc"}
    feats = tr.predict(gen.test_sequence(32))
    assert feats.shape == (64, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("n,q,class_order,with_qidx,C", [(3001, 37, True, True, 23), (3001, 37, False, True, 23), (4096, 64, True, False, 23),
                                                         (2048, 9, True, True, 23), (700, 300, True, True, 23), (5000, 40, True, True, 300),
                                                         (170001, 5, True, True, 23), (90000, 6, False, True, 700)])
def test_se_hierarchical_precision_general_rankings(n, q, class_order, with_qidx, C):
    """se_hierarchical_precision on rankings where the query sits ANYWHERE in its list (ranks ahead of it divide by the unshifted
    best curve, class_hierarchy.py:280-290), rows that are not 16-byte aligned (n = 3001), no query ids at all, cut-offs in several
    chunks, with and without the class-ordered visit, and all three sources of the gallery classes (byte table in LDS: C <= 256; 16-bit
    table: C = 300; global gather: tables that do not fit): equal to the NumPy statement of the reference's per-query loop (1e-10)."""
    import sehip
    from test_dp_gloo import _hprec_standin
    rng = np.random.default_rng(n + q)
    cls = rng.integers(0, C, size=n).astype(np.int32)
    tab_w = rng.random((C, C)) * 0.8 + 0.1; tab_w = (tab_w + tab_w.T) / 2; np.fill_diagonal(tab_w, 1.0)
    tab_l = rng.random((C, C)) * 0.8 + 0.1; tab_l = (tab_l + tab_l.T) / 2; np.fill_diagonal(tab_l, 1.0)
    counts = np.bincount(cls, minlength=C)

    def best(tab):
        return np.stack([np.cumsum(np.repeat(tab[c][np.argsort(-tab[c], kind="stable")], counts[np.argsort(-tab[c], kind="stable")]))
                         for c in range(C)])
    best_w, best_l = best(tab_w), best(tab_l)
    rk = np.stack([rng.permutation(n) for _ in range(q)]).astype(np.int32)
    rk[0, 0], rk[0, np.flatnonzero(rk[0] == 0)[0]] = 0, rk[0, 0]                      # query 0 is its own nearest neighbour
    last = np.flatnonzero(rk[1] == 1)[0]
    rk[1, last], rk[1, n - 1] = rk[1, n - 1], 1                                        # query 1 comes last in its list
    qidx = np.arange(q, dtype=np.int32)
    ks = np.array([250, 1, 2, 10, min(n - 1, 2047), min(n - 1, 2049), n - 1, 10], dtype=np.int32)      # unsorted, with a duplicate
    args = [torch.from_numpy(a) for a in (rk, cls, cls[:q].copy(), qidx if with_qidx else np.full(q, -1, np.int32), tab_w, tab_l, best_w, best_l, ks)]
    dev = [a.cuda() for a in args]
    if not with_qidx:
        dev[3] = None
    for ahp, ap in ((0, True), (50, False), (2500, True), (-1, False)):
        want = _hprec_standin(*args, ahp_len=ahp, want_ap=ap).numpy()
        got = sehip.hierarchical_precision(*dev, ahp_len=ahp, want_ap=ap, class_order=class_order).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-10, (ahp, ap, np.abs(got - want).max())
        if n <= 65536:
            # the same rankings as uint16 indices (se_hierarchical_precision_r16; int16 tensors hold the bit patterns): the SAME numbers
            # (only the width of the loads differs; rows that are not 16-byte aligned -- n = 3001 -- take the element loads)
            dev16 = [dev[0].to(torch.int16)] + dev[1:]
            got16 = sehip.hierarchical_precision(*dev16, ahp_len=ahp, want_ap=ap, class_order=class_order).cpu().numpy()
            assert np.array_equal(got16, got), (ahp, ap, np.abs(got16 - got).max())


@pytest.mark.gpu
@pytest.mark.parametrize("nk", [1, 250, 300, 512])
def test_se_hierarchical_precision_consecutive_cutoffs(nk):
    """ks = 1, 2, ..., K in order (what evaluate_retrieval.main asks for with --plot_max K): the kernel's slot == rank short cut, across a
    workgroup's ranking threads (K > 256), with the query at rank 0, in the middle of the cut-offs and absent."""
    import sehip
    from test_dp_gloo import _hprec_standin
    rng = np.random.default_rng(nk)
    n, q, C = 1500, 24, 31
    cls = rng.integers(0, C, size=n).astype(np.int32)
    tab = rng.random((C, C)) * 0.8 + 0.1; tab = (tab + tab.T) / 2; np.fill_diagonal(tab, 1.0)
    counts = np.bincount(cls, minlength=C)
    best = np.stack([np.cumsum(np.repeat(tab[c][np.argsort(-tab[c], kind="stable")], counts[np.argsort(-tab[c], kind="stable")])) for c in range(C)])
    rk = np.stack([rng.permutation(n) for _ in range(q)]).astype(np.int32)
    for r in range(q):          # query r at rank 0 (r even) or at rank 7 r (r odd); query 5 is not in its list at all
        want_pos = 0 if r % 2 == 0 else min(7 * r, n - 1)
        at = np.flatnonzero(rk[r] == r)[0]
        rk[r, at], rk[r, want_pos] = rk[r, want_pos], r
    qidx = np.arange(q, dtype=np.int32)
    qidx[5] = -1
    ks = np.arange(1, nk + 1, dtype=np.int32)
    args = [torch.from_numpy(a) for a in (rk, cls, cls[:q].copy(), qidx, tab, tab.T.copy(), best, best, ks)]
    dev = [a.cuda() for a in args]
    for ahp, ap in ((0, True), (100, False)):
        want = _hprec_standin(*args, ahp_len=ahp, want_ap=ap).numpy()
        got = sehip.hierarchical_precision(*dev, ahp_len=ahp, want_ap=ap).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-10, (ahp, ap, np.abs(got - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cub", "ilsvrc"])
def test_hierarchical_precision_device_other_taxonomies(name, tmp_path):
    """Device metrics on CUB (200 classes: byte class table) and the ILSVRC min-tree (1000 string-id classes: 16-bit class table when
    the lists are long, global gather for the clipped ones; queries without any relevant image) vs the REFERENCE's own outputs
    (tests/golden/hierarchy_{cub,ilsvrc}.npz), full rankings and the fused top-L path."""
    from test_host import TIE_NOISE, _hierarchy_from_fixture
    g = np.load(os.path.join(GOLDEN, "hierarchy_%s.npz" % name))
    h, labels = _hierarchy_from_fixture(g, tmp_path)
    ks = g["ks"].tolist()
    want = dict(zip(g["metric_names"].tolist(), g["metric_values"].tolist()))
    for norm in (True, False):
        for ahp in (True, 50):
            avg, per_q = h.hierarchical_precision_device(g["features"].copy(), labels, ks, compute_ahp=ahp, compute_ap=True, normalize=norm)
            assert len(per_q["AP"]) == len(labels)
            for m, v in avg.items():
                assert v == pytest.approx(want["%s|norm=%d|ahp=%s" % (m, norm, ahp)], rel=1e-10 + TIE_NOISE[norm], abs=1e-10 + TIE_NOISE[norm]), (m, norm, ahp)
        head, none = h.hierarchical_precision_device(g["features"].copy(), labels, ks, compute_ahp=50, compute_ap=False, normalize=norm, per_query=False)
        assert none is None
        for m, v in head.items():      # P@k and AHP@50 from top-101 lists only
            assert v == pytest.approx(want["%s|norm=%d|ahp=50" % (m, norm)], rel=1e-10 + TIE_NOISE[norm], abs=1e-10 + TIE_NOISE[norm]), (m, norm)


@pytest.mark.gpu
def test_training_graph_replay_matches_eager_steps_and_keeps_state():
    """Trainer.enable_graphs (fp32 NCHW backbone_mode of the CIFAR ResNets): capture must leave parameters, velocity and
    BatchNorm buffers untouched, replayed steps must follow the eager trajectory, a short batch runs eagerly."""
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer, backbone_mode
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    Ed = torch.from_numpy(E.astype(np.float32)).cuda()
    adt, fmt = backbone_mode("resnet-32")
    assert adt is None and fmt == torch.contiguous_format
    gen = SyntheticGenerator(100, 32, 3, 256, 32)
    seq = gen.train_sequence(32, shuffle=False)
    batches = [seq[i] for i in range(4)]

    def make():
        torch.manual_seed(0)
        model = utils.build_network(100, "resnet-32", classification=True, no_softmax=True, input_channels=3).cuda()
        l2 = {id(p): model.regularizer for p in model.regularized_parameters()}
        return Trainer(model, {"l2norm": (utils.CosineEmbeddingLoss(Ed), 1.0)}, {"l2norm": [utils.nn_accuracy(Ed, dot_prod_sim=True)]},
                       lr=0.05, clipnorm=10.0, l2_of=l2, autocast_dtype=adt, memory_format=fmt)

    eager, graph = make(), make()
    p0 = graph.flat.flat_p.clone()
    b0 = [b.clone() for b in graph.model.buffers()]
    assert graph.enable_graphs(*batches[0]) is True
    assert torch.equal(graph.flat.flat_p, p0) and float(graph.flat.flat_v.abs().max()) == 0.0 and graph.iterations == 0
    assert all(torch.equal(a, b) for a, b in zip(graph.model.buffers(), b0))
    le, lg = {}, {}
    for i in range(6):
        a = float(eager.train_step(*batches[i % 4], le))
        b = float(graph.train_step(*batches[i % 4], lg))
        assert np.isfinite(b) and abs(a - b) < 2e-3 * max(1.0, abs(a)), (i, a, b)
    rel = float(torch.linalg.vector_norm(eager.flat.flat_p - graph.flat.flat_p) / torch.linalg.vector_norm(eager.flat.flat_p))
    assert rel < 1e-3, rel
    assert le["_n"] == lg["_n"] == 6 * 32           # logs hold sums over samples; '_n' counts them (eager and replayed steps alike)
    assert abs(float(le["max_sim_acc"]) - float(lg["max_sim_acc"])) / le["_n"] < 1e-6 + 0.02
    assert abs(float(le["loss"]) - float(lg["loss"])) / le["_n"] < 2e-3
    Xs, ys = batches[0][0][:8], batches[0][1][:8]          # short batch: eager launches inside a graph-mode trainer
    assert np.isfinite(float(graph.train_step(Xs, ys, lg))) and graph.iterations == 7


def test_hierarchical_precision_device_matches_reference_values():
    """se_hierarchical_precision + device rankings vs the values the REFERENCE's class_hierarchy produced on its own
    rankings (tests/golden/hierarchy_cifar.npz), both branches, whole-list and clipped AHP, AP; per-query values also
    agree with the host mirror (float64: 1e-10)."""
    import tempfile
    from class_hierarchy import ClassHierarchy
    from oracle import retrieval_oracle as ro
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hierarchy_cifar.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for p, c in g["edges"]:
            f.write("%d %d\n" % (p, c))
    h = ClassHierarchy.from_file(f.name, id_type=int)
    os.unlink(f.name)
    labels = g["labels"].tolist()
    ks = g["ks"].tolist()
    want = dict(zip(g["metric_names"].tolist(), g["metric_values"].tolist()))
    for norm in (True, False):
        _, rk = ro.canon_retrieval(g["features"], norm)
        ret = {i: rk[i].tolist() for i in range(len(labels))}
        for ahp in (True, 50):
            avg, per_q = h.hierarchical_precision_device(g["features"].copy(), labels, ks, compute_ahp=ahp, compute_ap=True, normalize=norm)
            for m, v in avg.items():
                assert v == pytest.approx(want["%s|norm=%d|ahp=%s" % (m, norm, ahp)], rel=1e-10, abs=1e-10), (m, norm, ahp)
            _, host_q = h.hierarchical_precision(ret, labels, ks, compute_ahp=ahp, compute_ap=True, all_ids=list(range(len(labels))))
            for m in host_q:
                a = np.array([per_q[m][i] for i in range(len(labels))])
                b = np.array([host_q[m][i] for i in range(len(labels))])
                assert np.abs(a - b).max() <= 1e-10, m


def test_hierarchical_precision_device_multi_chunk_and_tiles():
    """Lists longer than one 2048-rank chunk, several query tiles, clustered classes: device == host mirror."""
    import tempfile
    from class_hierarchy import ClassHierarchy
    from oracle import retrieval_oracle as ro
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hierarchy_cifar.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for p, c in g["edges"]:
            f.write("%d %d\n" % (p, c))
    h = ClassHierarchy.from_file(f.name, id_type=int)
    os.unlink(f.name)
    rng = np.random.default_rng(12)
    n, d = 5000, 16
    labels = rng.integers(0, 100, size=n).tolist()
    centers = rng.standard_normal((100, d)).astype(np.float32)
    feats = (centers[labels] + 0.7 * rng.standard_normal((n, d))).astype(np.float32)
    ks = [1, 10, 50, 100, 250]
    avg, per_q = h.hierarchical_precision_device(feats.copy(), labels, ks, compute_ahp=True, compute_ap=True, normalize=True, tile_rows=1536)
    _, rk = ro.canon_retrieval(feats, True)
    sample = list(range(0, n, 97))
    ret = {i: rk[i].tolist() for i in sample}
    _, host_q = h.hierarchical_precision(ret, labels, ks, compute_ahp=True, compute_ap=True)
    for m in host_q:
        a = np.array([per_q[m][i] for i in sample])
        b = np.array([host_q[m][i] for i in sample])
        assert np.abs(a - b).max() <= 1e-10, m


def test_evaluate_retrieval_cli_end_to_end(tmp_path, capsys):
    """evaluate_retrieval.main with the reference's flags: feature pickle ({'feat': {id: vec}}) + hierarchy file +
    a synthetic dataset's test labels -> table of metrics; values equal the host mirror fed with oracle rankings."""
    import evaluate_retrieval as er
    from class_hierarchy import ClassHierarchy
    from datasets import get_data_generator
    g = np.load(os.path.join(GOLDEN, "hierarchy_cifar.npz"))
    hpath = tmp_path / "cifar.parent-child.txt"
    with open(hpath, "w") as f:
        for p, c in g["edges"]:
            f.write("%d %d\n" % (p, c))
    ds = "synthetic:100x8x64x300"
    gen = get_data_generator(ds, None)
    labels = list(gen.labels_test)
    rng = np.random.default_rng(4)
    centers = rng.standard_normal((100, 24)).astype(np.float32)
    feats = (centers[labels] + 0.8 * rng.standard_normal((len(labels), 24))).astype(np.float32)
    dump = tmp_path / "feat.pickle"
    with open(dump, "wb") as f:
        pickle.dump({"feat": {i: feats[i] for i in range(len(labels))}}, f)
    csv = tmp_path / "perf.csv"
    perf = er.main(["--dataset", ds, "--data_root", str(tmp_path), "--hierarchy", str(hpath), "--feat", str(dump), "--label", "run", "--norm", "yes",
                    "--plot_max", "0", "--csv", str(csv)])
    out = capsys.readouterr().out
    assert "AHP (WUP)" in out and os.path.exists(csv)
    h = ClassHierarchy.from_file(str(hpath), id_type=int)
    _, rk = ro.canon_retrieval(feats, True)
    want, _ = h.hierarchical_precision({i: rk[i].tolist() for i in range(len(labels))}, labels, [1, 10, 50, 100], compute_ahp=True,
                                       compute_ap=True, all_ids=list(range(len(labels))))
    for m, v in want.items():
        assert perf["run"][m] == pytest.approx(v, rel=1e-10, abs=1e-10), m


@pytest.mark.gpu
def test_learn_image_embeddings_cli_end_to_end(tmp_path, capsys):
    """The training CLI with the reference's flags on a synthetic dataset: two epochs of ResNet-110-fc against the CIFAR-100
    unit-sphere class embeddings (HIP loss head, HIP-graph replay of the step, validation, log, dumps), then the dumped
    features go through evaluate_retrieval.pairwise_retrieval."""
    import json
    import learn_image_embeddings as lie
    import evaluate_retrieval as er
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    emb = str(tmp_path / "emb.pickle")
    with open(emb, "wb") as f:
        pickle.dump({"embedding": E, "ind2label": list(range(100)), "label2ind": {i: i for i in range(100)}}, f)
    feat, wts, logd = str(tmp_path / "feat.pickle"), str(tmp_path / "w.pt"), str(tmp_path / "log")
    final = lie.main(["--dataset", "synthetic:100x32x192x64", "--data_root", "-", "--embedding", emb, "--architecture", "resnet-110-fc",
                      "--loss", "inv_corr", "--lr_schedule", "SGD", "--sgd_lr", "0.05", "--epochs", "2", "--batch_size", "32",
                      "--val_batch_size", "32", "--feature_dump", feat, "--weight_dump", wts, "--log_dir", logd, "--no_progress"])
    assert np.isfinite(final["loss"]) and 0.0 <= final["max_sim_acc"] <= 1.0, final
    log = [json.loads(l) for l in open(os.path.join(logd, "training_log.jsonl"))]
    assert [e["epoch"] for e in log] == [1, 2] and all(np.isfinite(e["loss"]) and np.isfinite(e["val_loss"]) for e in log)
    assert os.path.getsize(wts) > 1_000_000
    with open(feat, "rb") as f:
        dump = pickle.load(f)
    feats = np.stack([dump["feat"][i] for i in range(64)])
    assert feats.shape == (64, 100) and np.allclose(np.linalg.norm(feats, axis=-1), 1.0, atol=1e-4)      # the model ends in the l2norm layer
    ranked = dict(er.pairwise_retrieval(feat, normalize=True, return_generator=False))
    assert sorted(ranked) == list(range(64)) and all(ranked[i][0] == i and len(ranked[i]) == 64 for i in ranked)


# ---------------------------------------------------------------- world_size 2 on ONE GPU (gloo): the N > 1 graph-mode step

def _graph_dp_worker(rank, world, port, out):
    import torch.distributed as dist
    for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # RCCL refuses two ranks on one device; gloo all-reduces CUDA tensors
    torch.cuda.set_device(0)
    import utils
    from datasets import SyntheticGenerator
    from engine import Trainer, backbone_mode
    E = np.load(os.path.join(GOLDEN, "embeddings.npz"))["cifar100_unitsphere"]
    Ed = torch.from_numpy(E.astype(np.float32)).cuda()
    adt, fmt = backbone_mode("resnet-32")
    torch.manual_seed(0)
    model = utils.build_network(100, "resnet-32", classification=True, no_softmax=True, input_channels=3).cuda()
    tr = Trainer(model, {"l2norm": (utils.CosineEmbeddingLoss(Ed), 1.0)}, {"l2norm": [utils.nn_accuracy(Ed, dot_prod_sim=True)]},
                 lr=0.05, clipnorm=10.0, autocast_dtype=adt, memory_format=fmt)
    assert tr.world == 2 and tr.reducer.enabled
    gen = SyntheticGenerator(100, 32, 3, 256, 32)
    seq = gen.train_sequence(32, shuffle=False, rank=rank, world_size=world)
    ok = tr.enable_graphs(*seq[0])
    logs = {}
    losses = [float(tr.train_step(*seq[i % len(seq)], logs)) for i in range(4)]
    torch.cuda.synchronize()
    flat = tr.flat.flat_p.detach().cpu()
    both = [None, None]
    dist.all_gather_object(both, flat.numpy().tobytes())
    if rank == 0:
        torch.save({"ok": ok, "same": both[0] == both[1], "finite": bool(np.isfinite(losses).all()), "mode": tr._graph is not None}, out)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_training_graph_mode_world2_keeps_ranks_in_sync(tmp_path):
    """Two processes (gloo) on the one GPU: capture + agreement all-reduce + graph A | eager all-reduce | graph B must leave
    both ranks with bit-identical parameters, like the eager bucketed path does."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "g.pt")
    mp.spawn(_graph_dp_worker, args=(2, 29637, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["ok"] and got["mode"] and got["same"] and got["finite"], got


def test_nearest_centroid_classification_on_device():
    """evaluate_classification_accuracy.nn_classification (reference :51-71: cdist(feat, centroids, 'sqeuclidean').argsort)
    on se_pairwise_dist + se_rank_rows: equals SciPy's float64 ranking wherever adjacent class distances are clearly apart,
    equals the canonical oracle bit for bit, and feeds `evaluate` (flat / top-5 / balanced / hierarchical accuracy)."""
    import evaluate_classification_accuracy as eca
    from scipy.spatial.distance import cdist
    from oracle import retrieval_oracle as ro
    g = np.load(os.path.join(GOLDEN, "embeddings.npz"))
    E = g["cifar100_unitsphere"]
    rng = np.random.default_rng(4)
    y = rng.integers(0, 100, size=700)
    feats = (E[y] + 0.06 * rng.standard_normal((700, 100))).astype(np.float32)
    rank = eca.nn_classification(feats, {"embedding": E})
    assert rank.shape == (700, 100)
    want = ro.canon_rank_rows(ro.canon_pdist(feats, E.astype(np.float32), ro.METRIC_EUCLID))
    assert np.array_equal(rank, want)
    d64 = cdist(feats, E, "sqeuclidean")
    ref = d64.argsort(axis=-1)
    srt = np.sort(d64, axis=-1)
    clear = np.concatenate([np.ones((700, 1), bool), (srt[:, 1:] - srt[:, :-1]) > 1e-4], axis=1)
    clear &= np.concatenate([clear[:, 1:], np.ones((700, 1), bool)], axis=1)
    assert clear.mean() > 0.95 and np.array_equal(rank[clear], ref[clear])
    # device tensor in, device ranking out
    rk_dev = eca.nn_classification(torch.from_numpy(feats).cuda(), torch.from_numpy(E).cuda(), return_device=True)
    assert rk_dev.is_cuda and np.array_equal(rk_dev.cpu().numpy(), rank)

    class Data(object):
        labels_test = y.tolist()
        classes = list(range(100))

    import class_hierarchy as ch
    h = np.load(os.path.join(GOLDEN, "hierarchy_cifar.npz"))
    parents, children = {}, {}
    for p, c in h["edges"].tolist():
        parents.setdefault(c, []).append(p)
        children.setdefault(p, []).append(c)
    hier = ch.ClassHierarchy(parents, children)
    perf = eca.evaluate(rank, Data, hier)
    top1 = rank[:, 0]
    assert perf["Accuracy"] == pytest.approx(np.mean(top1 == y))
    assert perf["Top-5 Accuracy"] == pytest.approx(np.mean((rank[:, :5] == y[:, None]).any(1)))
    freq = np.bincount(y)
    assert perf["Avg. Accuracy"] == pytest.approx(((top1 == y) / freq[y]).sum() / len(freq))
    assert perf["Hierarchical Accuracy"] == pytest.approx(np.mean([1.0 - hier.lcs_height(int(a), int(b)) for a, b in zip(top1, y)]))
    assert perf["Accuracy"] > 0.9 and perf["Hierarchical Accuracy"] >= perf["Accuracy"]


def test_predict_keeps_features_on_the_device_for_the_metric_path(tmp_path):
    """Trainer.predict(..., to_host=False) -> ClassHierarchy.hierarchical_precision_device with no host hop; the
    values equal the pickle route (features through the host) exactly."""
    import utils
    import class_hierarchy as ch
    from datasets import get_data_generator
    from engine import Trainer
    gen = get_data_generator("synthetic:10x32x64x96", ".")
    torch.manual_seed(0)
    model = utils.build_network(16, "resnet-32", input_channels=3).cuda()   # pooled 64-d features
    tr = Trainer(model, {}, {}, autocast_dtype=None)
    seq = gen.test_sequence(32)
    f_dev = tr.predict(seq, to_host=False)
    f_host = tr.predict(seq)
    assert torch.is_tensor(f_dev) and f_dev.is_cuda and f_dev.dtype == torch.float32
    assert np.allclose(f_dev.cpu().numpy(), f_host, rtol=1e-4, atol=1e-5)      # (two forward passes: MIOpen kernels are not bit-reproducible)
    parents = {i: [100 + i // 5] for i in range(10)}
    parents.update({100: [200], 101: [200]})
    children = {}
    for c, ps in parents.items():
        for p in ps:
            children.setdefault(p, []).append(c)
    hier = ch.ClassHierarchy(parents, children)
    labels = gen.labels_test
    a, _ = hier.hierarchical_precision_device(f_dev, labels, [1, 5, 10], compute_ahp=True, compute_ap=True, normalize=True)
    b, _ = hier.hierarchical_precision_device(f_dev.cpu().numpy(), labels, [1, 5, 10], compute_ahp=True, compute_ap=True, normalize=True)
    assert a == b


def _sharded_topk_worker(rank, world, port, out, d=200, kblocks=None):
    import torch.distributed as dist
    for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # RCCL refuses two ranks on one device
    torch.cuda.set_device(0)
    import sehip
    import sharded_retrieval as sr
    rng = np.random.default_rng(0)
    gallery = rng.standard_normal((3001, d)).astype(np.float32)
    gallery[1500:1510] = gallery[0:10]                                # exact ties ACROSS the two shards
    queries = gallery[:300].copy()
    lo_, hi_ = sr.shard_bounds(len(gallery), world)[rank]
    g = torch.from_numpy(gallery[lo_:hi_]).cuda()
    q = torch.from_numpy(queries).cuda()
    sehip.normalize_rows_(g)
    sehip.normalize_rows_(q)
    dd, i = sr.sharded_topk(q, g, 251, lo_, metric=sehip.METRIC_COSINE, kblocks=kblocks)      # the real kernels: se_retrieve_topk + se_topk_merge
    # this rank's own shard once more, checked against the oracle here (if the merged result is ever wrong, this says which stage was)
    ld_, li_ = sehip.retrieve_topk(q, g, 251, metric=sehip.METRIC_COSINE, col_offset=lo_, kblocks=kblocks)
    wd_, wi_ = ro.canon_topk_rows(ro.canon_pdist(q.cpu().numpy(), g.cpu().numpy(), ro.METRIC_COSINE, kblocks=kblocks), 251, col_offset=lo_)
    local_ok = bool(np.array_equal(li_.cpu().numpy(), wi_) and np.array_equal(ld_.cpu().numpy(), wd_))
    both = [None] * world
    dist.all_gather_object(both, (i.cpu().numpy().tobytes(), local_ok))
    if rank == 0:
        np.savez(out, d=dd.cpu().numpy(), i=i.cpu().numpy(), gallery=gallery, queries=queries, same=both[0][0] == both[1][0],
                 local_ok=np.array([b[1] for b in both]))
    dist.destroy_process_group()


@pytest.mark.parametrize("d,kblocks,port", [(200, None, 29641), (555, [278, 277], 29643)])
def test_sharded_gallery_topk_two_processes_real_kernels(tmp_path, d, kblocks, port):
    """The north_star retrieval split with the HIP kernels on both ranks (two processes on the one GPU, gloo transport):
    per-shard se_retrieve_topk -> all-gather -> se_topk_merge == canonical top-k over the whole gallery, identical on both
    ranks, ties across shards broken by the GLOBAL index; D = 555 with the BLAS K-block list on every shard (BASELINE configs[4]'s
    arithmetic: the chain restarts per block)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "st.npz")
    mp.spawn(_sharded_topk_worker, args=(2, port, out, d, kblocks), nprocs=2, join=True)
    g = np.load(out)
    assert bool(g["same"])
    gal = ro.canon_normalize_rows(g["gallery"])
    qs = ro.canon_normalize_rows(g["queries"])
    wd, wi = ro.canon_topk_rows(ro.canon_pdist(qs, gal, ro.METRIC_COSINE, kblocks=kblocks), 251)
    bad = np.nonzero((g["i"] != wi).any(axis=1) | (g["d"] != wd).any(axis=1))[0]
    detail = ""
    if len(bad):
        r = int(bad[0]); c = int(np.nonzero((g["i"][r] != wi[r]) | (g["d"][r] != wd[r]))[0][0])
        detail = "%d rows differ; first: row %d col %d got (%r, %d) want (%r, %d); per-rank shard results ok: %s" % (
            len(bad), r, c, g["d"][r, c], g["i"][r, c], wd[r, c], wi[r, c], g["local_ok"].tolist())
    assert not len(bad), detail


@pytest.mark.parametrize("arch,loss", [("resnet-32", "inv_corr"), ("resnet-110-fc", "softmax_corr")])
def test_learn_image_embeddings_cli_with_classifier_head(tmp_path, arch, loss):
    """--cls_weight > 0 (reference cls_model, learn_image_embeddings.py:16-45,127-135): the classifier branch sits on the model that
    already ends in the l2norm (softmax) layer, also for architectures without a final Dense layer (resnet-32 emits its 64 pooled
    features); both outputs are trained, logged under the reference's output names and the dumped features are the
    normalised ones."""
    import json
    import learn_image_embeddings as lie
    dim = 64 if arch == "resnet-32" else 100
    rng = np.random.default_rng(0)
    E = rng.standard_normal((100, dim))
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    emb = str(tmp_path / "emb.pickle")
    with open(emb, "wb") as f:
        pickle.dump({"embedding": E, "ind2label": list(range(100)), "label2ind": {i: i for i in range(100)}}, f)
    feat, logd = str(tmp_path / "feat.pickle"), str(tmp_path / "log")
    final = lie.main(["--dataset", "synthetic:100x32x96x32", "--data_root", "-", "--embedding", emb, "--architecture", arch,
                      "--loss", loss, "--cls_weight", "0.1", "--lr_schedule", "SGD", "--sgd_lr", "0.05", "--epochs", "1",
                      "--batch_size", "32", "--val_batch_size", "32", "--feature_dump", feat, "--log_dir", logd, "--no_progress"])
    head = "l2norm" if loss == "inv_corr" else "softmax"
    assert np.isfinite(final["loss"]) and np.isfinite(final[head + "_loss"]) and np.isfinite(final["prob_loss"]), final
    assert 0.0 <= final["prob_acc"] <= 1.0
    with open(feat, "rb") as f:
        feats = np.stack(list(pickle.load(f)["feat"].values()))
    assert feats.shape == (32, dim)
    if loss == "inv_corr":
        assert np.allclose(np.linalg.norm(feats, axis=-1), 1.0, atol=1e-4)
    else:
        assert np.allclose(feats.sum(axis=-1), 1.0, atol=1e-4) and feats.min() >= 0


def test_cls_model_base_is_the_normalised_output():
    import utils
    import learn_image_embeddings as lie
    torch.manual_seed(0)
    net = utils.build_network(16, "resnet-32", input_channels=3).cuda()        # no Dense head: 64 pooled features
    m = lie.ClsModel(net, 10, head="l2norm", width=64).cuda().eval()
    x = torch.randn(4, 3, 32, 32, device="cuda")
    base, logits = m(x)
    raw = net(x).float()
    assert torch.allclose(base, raw / raw.norm(dim=-1, keepdim=True), atol=1e-6) and logits.shape == (4, 10)
    assert torch.allclose(logits, m.prob(m.bn(torch.relu(base))), atol=1e-6)


def test_labelembed_model_mirrors_the_reference_head():
    """learn_labelembedding.labelembed_model (reference :40-56): three outputs, identity-initialised label embeddings, out2 behind a
    stop-gradient, the loss output produced by the fused kernel and trainable end to end."""
    import utils
    import learn_labelembedding as ll
    torch.manual_seed(0)
    base = utils.build_network(32, "resnet-32", input_channels=3).cuda()          # pooled 64-d embedding (no Dense head)
    m = ll.labelembed_model(base, 10).cuda()
    x = torch.randn(16, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (16,), device="cuda")
    (X, Y), targets = ll.transform_inputs(x, y, 10)
    emb, out1, loss = m(X, Y)
    assert emb.shape == (16, 64) and out1.shape == (16, 10) and loss.shape == (16, 1) and targets["labelembed_loss"].shape == (16, 1)
    assert torch.equal(m.labelembeddings.weight, torch.eye(10, device="cuda"))
    out = m.embedding_bn(torch.relu(emb.float()))
    want = lo.labelembed_loss(out1.detach().cpu().numpy(), m.out2(out).detach().cpu().numpy(), torch.eye(10)[y.cpu()].numpy(), y.cpu().numpy())
    assert np.abs(loss.detach().cpu().numpy().ravel() - want).max() <= 1e-4
    loss.mean().backward()
    g = {n: p.grad for n, p in m.named_parameters()}
    assert g["prob.weight"].abs().sum() > 0 and g["out2.weight"].abs().sum() > 0 and g["labelembeddings.weight"].abs().sum() >= 0
    assert all(torch.isfinite(v).all() for v in g.values() if v is not None)


def test_bench_line_schema_on_a_small_problem():
    """bench.py end to end at a reduced size (so that a broken bench cannot reach the driver): one JSON line with the contract's
    keys, the roofline / kernels objects, the post-run oracle verification and the sharded-gallery leg."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "4096", "--steps", "2", "--warmup", "1", "--no-train",
                          "--shard-rows", "3000", "--shard-queries", "500", "--shard-dim", "64", "--shard-k", "17", "--cpu-sample-queries", "256"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "kernels", "verified", "sharded_gallery", "hierarchical_precision"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["verified"] is True and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["kernels"]["pairwise_dist"]["flops_executed"] < d["kernels"]["pairwise_dist"]["flops_full_matrix"]
    hp = d["hierarchical_precision"]
    assert "error" not in hp and hp["finite"] and hp["queries"] == 4096 and hp["ms"] > 0
    sg = d["sharded_gallery"]
    assert "error" not in sg and sg["merged_lists_sorted_with_index_tiebreak"] and sg["merged_indices_in_range"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_cifar_pipeline_on_the_device_matches_reference_and_augmentation_statistics(tmp_path):
    """The in-memory data path on the GPU: un-augmented batches == the reference generator's (tests/golden/cifar_pipeline.npz,
    produced by the reference's unmodified datasets/cifar.py + datasets/common.py), and the device augmentation draws what
    ImageDataGenerator(horizontal_flip, width/height_shift_range = 0.15) draws -- flips with probability 1/2, shifts uniform in
    +-4.8 pixels -- and applies them like Keras (replaying the drawn parameters through apply_transform reproduces the batch)."""
    import pickle
    from datasets.cifar import CifarGenerator
    g = np.load(os.path.join(ROOT, "tests", "golden", "cifar_pipeline.npz"))
    for name, raw, lab in (("train", g["raw_train"], g["y_train"]), ("test", g["raw_test"], g["y_test"])):
        with open(os.path.join(str(tmp_path), name), "wb") as f:
            pickle.dump({b"data": raw, b"fine_labels": lab.tolist()}, f)
    gen = CifarGenerator(str(tmp_path))
    X, y = gen.test_sequence(batch_size=64)[0]
    assert X.is_cuda and np.array_equal(y.cpu().numpy(), g["test_y"])
    assert np.abs(X.permute(0, 2, 3, 1).cpu().numpy() - g["test_X"]).max() < 2e-6
    idx = np.arange(240).repeat(40)                                   # 9600 draws
    xb, (row, col, flip) = gen.compose_batch(idx, train=True, augment=True, return_params=True)
    assert abs(float(flip.float().mean()) - 0.5) < 0.03
    for t in (row, col):
        assert float(t.abs().max()) <= 0.15 * 32 + 1e-4 and abs(float(t.mean())) < 0.15 and abs(float(t.var()) - (9.6 ** 2) / 12.0) < 0.5
    plain = gen.compose_batch(idx, train=True, augment=False)
    assert torch.equal(gen.apply_transform(plain.contiguous(), row, col, flip).contiguous(memory_format=torch.channels_last), xb)
    # replaying the reference's drawn parameters on the device
    p = g["aug_params"]
    n = len(g["aug_X"])
    got = gen.apply_transform(gen.compose_batch(np.arange(n), train=True, augment=False).contiguous(), torch.from_numpy(p[:, 0]).float().cuda(),
                              torch.from_numpy(p[:, 1]).float().cuda(), torch.from_numpy(p[:, 2] != 0).cuda())
    assert np.abs(got.permute(0, 2, 3, 1).cpu().numpy() - g["aug_X"]).max() < 2e-4


def test_entry_points_under_two_stream_contention():
    """tools/stress_streams.py (short form): every retrieval / loss entry point, called while a second stream keeps the CUs busy,
    returns what it returns alone, bit for bit (round 4: se_topk_rows did not -- DESIGN.md section 5.6)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_streams.py"), "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "stress_streams: 0 differing calls" in r.stdout
