#!/usr/bin/env python
"""ONE process, two streams: se_topk_rows (barrier-dense exact select + in-LDS bitonic sort) on one stream while another stream keeps the
GPU busy with distance kernels -- the same contention as tools/stress_two_procs.py but without a second process (no compute-wave
save / restore between processes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
    sys.path.insert(0, p)
import sehip
from oracle import retrieval_oracle as ro
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 251
rng = np.random.default_rng(0)
gallery = rng.standard_normal((3001, 200)).astype(np.float32)
gh = ro.canon_normalize_rows(gallery[:1501]); qh = ro.canon_normalize_rows(gallery[:300])
want_pd = ro.canon_pdist(qh, gh, 0)
wd, wi = ro.canon_topk_rows(want_pd, K)
pdg = torch.from_numpy(want_pd).cuda()
big = torch.from_numpy(rng.standard_normal((6000, 200)).astype(np.float32)).cuda()
side = torch.cuda.Stream()
bad = 0
for it in range(iters):
    with torch.cuda.stream(side):
        for _ in range(3):
            if mode in ("both", "pdist"):
                sehip.pairwise_dist(big, big, metric=0)
            if mode in ("both", "topk"):
                sehip.topk_rows(pdg, K)
            if mode == "copy":
                big2 = big.clone(); big2.mul_(1.0001)
    d2, i2 = sehip.topk_rows(pdg, K)
    if not (np.array_equal(i2.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)):
        bad += 1
torch.cuda.synchronize()
print("one process, two streams, side work = %s, k = %d: %d iterations, se_topk_rows wrong %d times" % (mode, K, iters, bad))
