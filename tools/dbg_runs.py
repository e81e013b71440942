import os, sys
sys.path[:0] = ["semantic-embeddings_amd", "."]
import numpy as np, torch, sehip
from oracle import retrieval_oracle as ro
for n in (60000, 100000):
    rng = np.random.default_rng(n)
    pd = rng.standard_normal((3, n)).astype(np.float32)
    got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()
    w = ro.canon_rank_rows(pd)
    for r in range(3):
        bad = np.nonzero(got[r] != w[r])[0]
        print(n, "row", r, "bad", len(bad), bad[:8], got[r][bad[:8]], w[r][bad[:8]])
        # is it a permutation? sorted?
        print("   perm", len(np.unique(got[r])) == n, "in range", got[r].min(), got[r].max(), "sorted keys", bool((np.diff(pd[r][np.clip(got[r],0,n-1)]) >= 0).all()))
