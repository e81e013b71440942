"""CPU, world_size = 2 over gloo: the data-parallel gradient reduction of engine.Trainer and the
sharded-gallery top-k exchange of sharded_retrieval (the N > 1 paths bench.py runs over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))


def _loss(y, o):
    return ((o - y) ** 2).sum(-1)


def _dp_worker(rank, world, port, out):
    _setup(rank, world, port)
    import engine
    model = _make_model()
    l2 = {id(model[0].weight): 1e-3}
    tr = engine.Trainer(model, {"o": (_loss, 1.0)}, {}, lr=0.05, momentum=0.9, clipnorm=1.0, l2_of=l2, autocast_dtype=None,
                        bucket_bytes=64)   # tiny buckets -> several async all-reduces per step
    assert tr.reducer.enabled and len(tr.reducer.buckets) > 1
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 4, generator=g)
    logs = {}
    for _ in range(4):
        tr.train_step(X[rank::world], Y[rank::world], logs)
    red = tr._reduce_logs(logs)
    if rank == 0:     # (the reported loss carries the L2 penalty of the current weights, like Keras' does)
        torch.save({"state": model.state_dict(), "loss": red["loss"] - tr.regularization_loss(), "reg": tr.regularization_loss()}, out)
    dist.destroy_process_group()


def test_dp_matches_single_process_on_the_concatenated_batch(tmp_path):
    import engine
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, 29611, out), nprocs=2, join=True)
    got = torch.load(out)
    model = _make_model()
    tr = engine.Trainer(model, {"o": (_loss, 1.0)}, {}, lr=0.05, momentum=0.9, clipnorm=1.0, l2_of={id(model[0].weight): 1e-3},
                        autocast_dtype=None)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 4, generator=g)
    logs = {}
    for _ in range(4):
        tr.train_step(X, Y, logs)
    for k, v in model.state_dict().items():
        assert torch.allclose(v, got["state"][k], atol=1e-6), k
    assert float(logs["loss"]) / float(logs["_n"]) == pytest.approx(got["loss"], rel=1e-5)
    assert got["reg"] == pytest.approx(1e-3 * float((model[0].weight ** 2).sum()), rel=1e-5) and got["reg"] > 0


def _topk_worker(rank, world, port, out):
    _setup(rank, world, port)
    import sharded_retrieval as sr
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(0)
    gallery = rng.standard_normal((101, 16)).astype(np.float32)
    gallery[50:60] = gallery[0:10]                      # exact ties across shards
    queries = gallery[:12].copy()
    lo, hi = sr.shard_bounds(len(gallery), world)[rank]

    def local_topk(q, g, k, off):   # CPU stand-in for sehip.retrieve_topk
        d, i = ro.canon_topk_rows(ro.canon_pdist(q.numpy(), g.numpy(), ro.METRIC_COSINE), k, col_offset=off)
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(d, i):                # CPU stand-in for sehip.topk_merge
        md, mi = ro.canon_topk_merge(d.numpy(), i.numpy())
        return torch.from_numpy(md), torch.from_numpy(mi)

    d, i = sr.sharded_topk(torch.from_numpy(queries), torch.from_numpy(gallery[lo:hi]), 9, lo, local_topk=local_topk, merge=merge)
    gathered = [None] * world
    dist.all_gather_object(gathered, i.numpy().tolist())
    assert gathered[0] == gathered[1]                   # identical on every rank
    if rank == 0:
        np.savez(out, d=d.numpy(), i=i.numpy(), gallery=gallery, queries=queries)
    dist.destroy_process_group()


def _packed_worker(rank, world, port, out):
    _setup(rank, world, port)
    import sharded_retrieval as sr
    q, k = 5, 7
    d = torch.full((q, k), float(rank) + 0.5) + torch.arange(k, dtype=torch.float32)
    i = torch.arange(q * k, dtype=torch.int32).view(q, k) + 1000 * rank
    g = sr.all_gather_packed(sr.pack_lists(d, i), world)
    assert tuple(g.shape) == (world, 2, q, k) and g.dtype == torch.int32 and g.is_contiguous()
    for r in range(world):                                   # rank-major blocks: distance bits, then indices
        assert torch.equal(g[r, 0].view(torch.float32), torch.full((q, k), float(r) + 0.5) + torch.arange(k, dtype=torch.float32))
        assert torch.equal(g[r, 1], torch.arange(q * k, dtype=torch.int32).view(q, k) + 1000 * r)
    if rank == 0:
        np.savez(out, ok=np.ones(1))
    dist.destroy_process_group()


def test_packed_all_gather_layout(tmp_path):
    """ONE all-gather of every rank's packed (distance bits | indices) block -> [world, 2, Q, k], the layout se_topk_merge_packed reads."""
    out = str(tmp_path / "packed.npz")
    mp.spawn(_packed_worker, args=(2, 29641, out), nprocs=2, join=True)
    assert os.path.exists(out)


def test_sharded_gallery_topk_equals_unsharded(tmp_path):
    from oracle import retrieval_oracle as ro
    out = str(tmp_path / "topk.npz")
    mp.spawn(_topk_worker, args=(2, 29613, out), nprocs=2, join=True)
    g = np.load(out)
    wd, wi = ro.canon_topk_rows(ro.canon_pdist(g["queries"], g["gallery"], ro.METRIC_COSINE), 9)
    assert np.array_equal(g["i"], wi)
    assert np.array_equal(g["d"], wd)


def _topk_kblocks_worker(rank, world, port, out):
    """Sharded-gallery top-k on a D = 1000 problem WITH the BLAS K-block list (BASELINE configs[4] is D = 1000): the list
    must reach every rank's local kernel, otherwise near-tie orders differ from the single-process / reference result."""
    _setup(rank, world, port)
    import sharded_retrieval as sr
    from oracle import retrieval_oracle as ro
    feat, norm, kb, head = ro.load_topk_fixture(os.path.join(ROOT, "tests", "golden", "topk_head_d1000_cos.npz"))
    x = ro.canon_normalize_rows(feat)
    lo, hi = sr.shard_bounds(len(x), world)[rank]
    seen = []

    def local_topk(q, g, k, off, kblocks=None):   # CPU stand-in for sehip.retrieve_topk(..., kblocks=)
        seen.append(kblocks)
        d, i = ro.canon_topk_rows(ro.canon_pdist(q.numpy(), g.numpy(), ro.METRIC_COSINE, kblocks=kblocks), k, col_offset=off)
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(d, i):
        md, mi = ro.canon_topk_merge(d.numpy(), i.numpy())
        return torch.from_numpy(md), torch.from_numpy(mi)

    d, i = sr.sharded_topk(torch.from_numpy(x), torch.from_numpy(x[lo:hi]), 251, lo, local_topk=local_topk, merge=merge, kblocks=kb)
    assert seen == [kb]
    if rank == 0:
        np.savez(out, d=d.numpy(), i=i.numpy())
    dist.destroy_process_group()


def test_sharded_gallery_topk_with_kblocks_equals_single_process_and_reference(tmp_path):
    from oracle import retrieval_oracle as ro
    out = str(tmp_path / "topk_kb.npz")
    mp.spawn(_topk_kblocks_worker, args=(2, 29621, out), nprocs=2, join=True)
    got = np.load(out)
    feat, norm, kb, head = ro.load_topk_fixture(os.path.join(ROOT, "tests", "golden", "topk_head_d1000_cos.npz"))
    pd = ro.canon_pdist(ro.canon_normalize_rows(feat), None, ro.METRIC_COSINE, kblocks=kb)
    wd, wi = ro.canon_topk_rows(pd, 251)
    assert np.array_equal(got["i"], wi) and np.array_equal(got["d"], wd)          # == one process with the same K-blocks
    for r in np.nonzero((got["i"] != head[:, :251]).any(axis=1))[0]:                # == the reference's head outside exact ties
        assert np.array_equal(pd[r][got["i"][r]], pd[r][head[r, :251]]), r
    # without the list the arithmetic is a different one on this fixture (this is what used to be dropped silently)
    _, wi1 = ro.canon_topk_rows(ro.canon_pdist(ro.canon_normalize_rows(feat), None, ro.METRIC_COSINE), 251)
    assert not np.array_equal(wi1, wi)


def test_hierarchical_precision_device_hands_kblocks_to_the_topk_path():
    """`hierarchical_precision_device(..., kblocks=...)` with head-only metrics takes the top-L path and must pass the
    K-block list on (it used to drop it whenever world > 1)."""
    from class_hierarchy import ClassHierarchy
    g = np.load(os.path.join(ROOT, "tests", "golden", "hierarchy_cifar.npz"))
    parents, children = {}, {}
    for p, c in g["edges"].tolist():
        parents.setdefault(c, []).append(p)
        children.setdefault(p, []).append(c)
    hier = ClassHierarchy(parents, children)
    labels, feats = g["labels"].tolist(), g["features"]
    kern = _cpu_kernels()
    seen = []
    inner = kern["local_topk"]

    def local_topk(q, g_, k, off, kblocks=None):
        seen.append(kblocks)
        return inner(q, g_, k, off)
    kern["local_topk"] = local_topk
    d = feats.shape[1]
    hier.hierarchical_precision_device(feats.copy(), labels, [1, 10], compute_ahp=50, compute_ap=False, normalize=True,
                                       kernels=kern, kblocks=[d - 40, 40])
    assert seen == [[d - 40, 40]]


def test_shard_bounds_cover_everything():
    import sharded_retrieval as sr
    for n, w in ((10, 3), (1281167, 8), (5, 8)):
        b = sr.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


# ---------------------------------------------------------------- rank-aware retrieval evaluation

def _hprec_standin(rank_tile, cls, qcls, qidx, wup, lcs, best_w, best_l, ks, ahp_len=-1, want_ap=False, list_len=None):
    """NumPy stand-in for sehip.hierarchical_precision (se_hierarchical_precision's contract, class_hierarchy.py:257-314)."""
    trapz = getattr(np, "trapezoid", None) or np.trapz
    rank_tile, cls, qcls, qidx = (t.numpy() for t in (rank_tile, cls, qcls, qidx))
    wup, lcs, best_w, best_l, ks = (t.numpy() for t in (wup, lcs, best_w, best_l, ks))
    nk = len(ks)
    out = np.zeros((rank_tile.shape[0], 2 * nk + 3))
    for r in range(rank_tile.shape[0]):
        ret = rank_tile[r]
        cols = cls[ret]
        qp = np.nonzero(ret == qidx[r])[0]
        cw_best, cl_best = best_w[qcls[r]], best_l[qcls[r]]
        w, l = wup[qcls[r], cols], lcs[qcls[r], cols]
        rel = cols == qcls[r]
        if len(qp):
            p = int(qp[0])
            w, l, rel = np.delete(w, p), np.delete(l, p), np.delete(rel, p)
            cw_best = np.concatenate((cw_best[:p], cw_best[p + 1:] - 1.0))
            cl_best = np.concatenate((cl_best[:p], cl_best[p + 1:] - 1.0))
        cw, cl = np.cumsum(w), np.cumsum(l)
        for t, k in enumerate(ks):
            kk = min(int(k), len(cw))
            out[r, t] = cw[kk - 1] / cw_best[k - 1]
            out[r, nk + t] = cl[kk - 1] / cl_best[k - 1]
        if ahp_len == 0:
            out[r, 2 * nk] = trapz(cw / cw_best[:len(cw)], dx=1. / len(cw))
            out[r, 2 * nk + 1] = trapz(cl / cl_best[:len(cl)], dx=1. / len(cl))
        elif ahp_len > 0:
            out[r, 2 * nk] = trapz(cw[:ahp_len] / cw_best[:ahp_len], dx=1. / ahp_len)
            out[r, 2 * nk + 1] = trapz(cl[:ahp_len] / cl_best[:ahp_len], dx=1. / ahp_len)
        if want_ap:
            hits = np.flatnonzero(rel)
            out[r, 2 * nk + 2] = float(np.mean(np.arange(1, hits.size + 1) / (hits + 1.0))) if hits.size else 0.0
    return torch.from_numpy(out)


def _cpu_kernels():
    from oracle import retrieval_oracle as ro

    def ranking_tiles(features, normalize=False, tile_rows=None, idx64=False, queries=None, kblocks=None):
        f = features.numpy()
        if normalize:
            f[:] = ro.canon_normalize_rows(f)
        q0, q1 = (0, len(f)) if queries is None else queries
        step = 37                                          # several ragged tiles per shard
        for r0 in range(q0, q1, step):
            r1 = min(q1, r0 + step)
            pd = ro.canon_pdist(f[r0:r1], f, ro.METRIC_COSINE if normalize else ro.METRIC_EUCLID)
            yield r0, torch.from_numpy(ro.canon_rank_rows(pd))

    def local_topk(q, g, k, off):
        d, i = ro.canon_topk_rows(ro.canon_pdist(ro.canon_normalize_rows(q.numpy()), ro.canon_normalize_rows(g.numpy()), ro.METRIC_COSINE),
                                  k, col_offset=off)
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(d, i):
        md, mi = ro.canon_topk_merge(d.numpy(), i.numpy())
        return torch.from_numpy(md), torch.from_numpy(mi)

    return {"ranking_tiles": ranking_tiles, "hierarchical_precision": _hprec_standin, "local_topk": local_topk, "merge": merge,
            "device": torch.device("cpu")}


def _hprec_worker(rank, world, port, out):
    _setup(rank, world, port)
    from class_hierarchy import ClassHierarchy
    g = np.load(os.path.join(ROOT, "tests", "golden", "hierarchy_cifar.npz"))
    parents, children = {}, {}
    for p, c in g["edges"].tolist():
        parents.setdefault(c, []).append(p)
        children.setdefault(p, []).append(c)
    hier = ClassHierarchy(parents, children)
    labels, feats, ks = g["labels"].tolist(), g["features"], g["ks"].tolist()
    res = {}
    # full rankings (AP + un-clipped AHP): queries sharded, per-query rows all-gathered
    res["full"], per_query = hier.hierarchical_precision_device(feats.copy(), labels, ks, compute_ahp=True, compute_ap=True, normalize=True,
                                                               distributed=True, kernels=_cpu_kernels())
    assert len(per_query["AP"]) == len(labels)
    # the same with only the sums reduced
    res["sums"], local = hier.hierarchical_precision_device(feats.copy(), labels, ks, compute_ahp=True, compute_ap=True, normalize=True,
                                                            distributed=True, gather_per_query=False, kernels=_cpu_kernels())
    assert 0 < len(local["AP"]) < len(labels)
    # head-only metrics: gallery sharded, all-gather of per-shard top-k + merge
    res["head"], _ = hier.hierarchical_precision_device(feats.copy(), labels, ks, compute_ahp=50, compute_ap=False, normalize=True,
                                                        distributed=True, kernels=_cpu_kernels())
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    for name in res:
        for m in res[name]:
            assert gathered[0][name][m] == pytest.approx(gathered[1][name][m], rel=1e-13), (name, m)
    if rank == 0:
        import json
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


def test_rank_aware_hierarchical_precision_equals_reference_outputs(tmp_path):
    """world 2 (gloo): query-sharded full rankings and sharded-gallery top-k both reproduce the values the imported
    reference produced on one process (tests/golden/hierarchy_cifar.npz: ClassHierarchy.hierarchical_precision over
    evaluate_retrieval.pairwise_retrieval, class_hierarchy.py:211-316)."""
    import json
    out = str(tmp_path / "hprec.json")
    mp.spawn(_hprec_worker, args=(2, 29617, out), nprocs=2, join=True)
    with open(out) as f:
        res = json.load(f)
    g = np.load(os.path.join(ROOT, "tests", "golden", "hierarchy_cifar.npz"))
    ref = dict(zip(g["metric_names"].tolist(), g["metric_values"].tolist()))
    checked = 0
    for m, v in res["full"].items():
        assert v == pytest.approx(ref["%s|norm=1|ahp=True" % m], abs=1e-10), m
        assert res["sums"][m] == pytest.approx(v, abs=1e-12), m
        checked += 1
    for m, v in res["head"].items():
        assert v == pytest.approx(ref["%s|norm=1|ahp=50" % m], abs=1e-10), m
        checked += 1
    assert checked >= 2 * (2 * len(g["ks"]) + 2)


def test_evaluate_retrieval_main_is_rank_aware(monkeypatch):
    """`evaluate_retrieval.init_distributed` follows the torch.distributed.run environment, and `main` hands `distributed`
    to the device metric path (checked without a GPU by stubbing the heavy pieces)."""
    import evaluate_retrieval as er
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert er.init_distributed() == (0, 1)
    src = open(er.__file__).read()
    assert "distributed=world > 1" in src and "if rank != 0:" in src


def _two_trainers_worker(rank, world, port, out):
    """The --finetune flow (learn_image_embeddings.py:183-207): a first trainer on the new layers only, then a second one on the whole
    model.  The first one's gradient hooks must be gone (Trainer.close) -- otherwise they all-reduce a dead buffer on every backward
    of the second -- and uneven shards (3 + 2 rows) must be averaged per SAMPLE."""
    _setup(rank, world, port)
    import engine
    model = _make_model()
    g = torch.Generator().manual_seed(2)
    X, Y = torch.randn(5, 6, generator=g), torch.randn(5, 4, generator=g)
    pre = engine.Trainer(model, {"o": (_loss, 1.0)}, {}, lr=0.05, momentum=0.0, autocast_dtype=None, trainable=lambda n: n.startswith("2."))
    pre.train_step(X[rank::world], Y[rank::world], {})
    pre.close()
    assert pre.reducer.hook_handles == [] and not pre.reducer.enabled
    for p in model.parameters():
        p.requires_grad_(True)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(t.data_ptr()), orig(t, *a, **k))[1]
    tr = engine.Trainer(model, {"o": (_loss, 1.0)}, {}, lr=0.05, momentum=0.0, autocast_dtype=None)
    logs = {}
    tr.train_step(X[rank::world], Y[rank::world], logs)
    dist.all_reduce = orig
    lo_, hi_ = tr.flat.flat_g.data_ptr(), tr.flat.flat_g.data_ptr() + tr.flat.total * 4
    assert calls and all(lo_ <= c < hi_ for c in calls), "an all-reduce touched a buffer of the closed trainer"
    red = tr._reduce_logs(logs)
    if rank == 0:
        torch.save({"loss": red["loss"], "n": float(logs["_n"])}, out)
    dist.destroy_process_group()


def test_second_trainer_after_close_and_per_sample_log_weighting(tmp_path):
    out = str(tmp_path / "two.pt")
    mp.spawn(_two_trainers_worker, args=(2, 29619, out), nprocs=2, join=True)
    got = torch.load(out)
    import engine
    model = _make_model()
    g = torch.Generator().manual_seed(2)
    X, Y = torch.randn(5, 6, generator=g), torch.randn(5, 4, generator=g)
    pre = engine.Trainer(model, {"o": (_loss, 1.0)}, {}, lr=0.05, momentum=0.0, autocast_dtype=None, trainable=lambda n: n.startswith("2."))
    # the data-parallel pre-step averages the two shard MEANS (3 and 2 rows), like every DP step does; reproduce that weighting here
    o = model(X)
    per = _loss(Y, o)
    (0.5 * (per[0::2].mean() + per[1::2].mean())).backward()
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.startswith("2."):
                p -= 0.05 * p.grad
            p.grad = None
    want = float(_loss(Y, model(X)).mean())            # per-SAMPLE mean over all five rows
    assert got["n"] == 3.0 and got["loss"] == pytest.approx(want, rel=1e-5)
