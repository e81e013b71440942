import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd")); sys.path.insert(0, ROOT)
import sehip
from oracle import retrieval_oracle as ro
for n in (4097, 1000, 5000, 50000):
    rng = np.random.default_rng(n)
    pd = rng.standard_normal((9, n)).astype(np.float32)
    got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()
    want = ro.canon_rank_rows(pd)
    bad = np.argwhere(got != want)
    print(n, "mismatches", len(bad))
    if len(bad):
        r, c = bad[0]
        print(" first at row", r, "pos", c, "got", got[r, c-2:c+6], "want", want[r, c-2:c+6])
        print(" keys got", pd[r, got[r, c-2:c+6]], "want", pd[r, want[r, c-2:c+6]])
        print(" is permutation:", np.array_equal(np.sort(got[r]), np.arange(n)), " sorted by value:", bool(np.all(np.diff(pd[r, got[r]]) >= 0)))
        cols = np.unique(bad[:, 1]); print(" bad positions range", cols.min(), cols.max(), "count", len(cols))
