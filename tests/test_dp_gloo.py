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
    red = tr._reduce_logs(logs, 4)
    if rank == 0:
        torch.save({"state": model.state_dict(), "loss": red["loss"]}, out)
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
    assert float(logs["loss"]) / 4 == pytest.approx(got["loss"], rel=1e-5)


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


def test_sharded_gallery_topk_equals_unsharded(tmp_path):
    from oracle import retrieval_oracle as ro
    out = str(tmp_path / "topk.npz")
    mp.spawn(_topk_worker, args=(2, 29613, out), nprocs=2, join=True)
    g = np.load(out)
    wd, wi = ro.canon_topk_rows(ro.canon_pdist(g["queries"], g["gallery"], ro.METRIC_COSINE), 9)
    assert np.array_equal(g["i"], wi)
    assert np.array_equal(g["d"], wd)


def test_shard_bounds_cover_everything():
    import sharded_retrieval as sr
    for n, w in ((10, 3), (1281167, 8), (5, 8)):
        b = sr.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
