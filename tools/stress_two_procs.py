#!/usr/bin/env python
"""Two processes on ONE GPU hammering the slab path of se_retrieve_topk (se_pairwise_dist + se_topk_rows) and se_normalize_rows with
barriers in between, every result compared with the oracle: which kernel gives the rare wrong row of
tests/test_gpu_dropin.py::test_sharded_gallery_topk_two_processes_real_kernels?"""
import os, sys
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, iters):
    for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import sehip
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(0)
    gallery = rng.standard_normal((3001, 200)).astype(np.float32)
    lo, hi = (0, 1501) if rank == 0 else (1501, 3001)
    gh = ro.canon_normalize_rows(gallery[lo:hi]); qh = ro.canon_normalize_rows(gallery[:300])
    want_pd = ro.canon_pdist(qh, gh, 0)
    wd, wi = ro.canon_topk_rows(want_pd, 251, col_offset=lo)
    bad = {"norm": 0, "pdist": 0, "topk_rows": 0, "retrieve": 0}
    for it in range(iters):
        dist.barrier()
        g = torch.from_numpy(gallery[lo:hi]).cuda(); q = torch.from_numpy(gallery[:300]).cuda()
        sehip.normalize_rows_(g); sehip.normalize_rows_(q)
        if not (np.array_equal(g.cpu().numpy(), gh) and np.array_equal(q.cpu().numpy(), qh)): bad["norm"] += 1
        g = torch.from_numpy(gh).cuda(); q = torch.from_numpy(qh).cuda()
        pd = sehip.pairwise_dist(q, g, metric=0)
        pdh = pd.cpu().numpy()
        if not np.array_equal(pdh, want_pd):
            bad["pdist"] += 1
            r, c = np.nonzero(pdh != want_pd)
            print("rank %d it %d: pdist wrong at %d positions, first (%d, %d): got %r want %r" % (rank, it, len(r), r[0], c[0], pdh[r[0], c[0]], want_pd[r[0], c[0]]), flush=True)
        pdg = torch.from_numpy(want_pd).cuda()
        d2, i2 = sehip.topk_rows(pdg, 251, col_offset=lo)
        if not (np.array_equal(i2.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)):
            bad["topk_rows"] += 1
            gi, gd = i2.cpu().numpy(), d2.cpu().numpy()
            rr = np.nonzero((gi != wi).any(axis=1))[0]
            r0 = int(rr[0]); c0 = int(np.nonzero(gi[r0] != wi[r0])[0][0])
            missing = sorted(set(wi[r0].tolist()) - set(gi[r0].tolist())); extra = sorted(set(gi[r0].tolist()) - set(wi[r0].tolist()))
            srt = bool((np.diff(gd[r0]) >= 0).all())
            print("rank %d it %d: topk_rows wrong in rows %s; row %d first diff col %d: got (%r, %d) want (%r, %d); missing %s extra %s sorted %s; want key of missing %s" %
                  (rank, it, rr[:5], r0, c0, gd[r0, c0], gi[r0, c0], wd[r0, c0], wi[r0, c0], missing[:4], extra[:4], srt,
                   [float(want_pd[r0, m - lo]) for m in missing[:4]]), flush=True)
        d3, i3 = sehip.retrieve_topk(q, g, 251, metric=0, col_offset=lo)
        if not (np.array_equal(i3.cpu().numpy(), wi) and np.array_equal(d3.cpu().numpy(), wd)): bad["retrieve"] += 1
    print("rank %d: %d iterations, failures %s" % (rank, iters, bad), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29677, int(sys.argv[1]) if len(sys.argv) > 1 else 60), nprocs=2, join=True)
