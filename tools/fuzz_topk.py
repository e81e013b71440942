"""Randomised check of se_retrieve_topk / se_topk_merge against the canonical oracle (oracle/canon.c), aimed at the round-4 kernels:
the fp16 pre-filter with both tile kernels (128 x 128; 256 x 256 LDS-DMA from a padded width of 256), the refinement with direct and
LDS-staged row gathers, the exact fallback (duplicate-heavy galleries overflow their lists; NaN / huge rows are irregular), K-blocks,
column offsets, and the wave-per-query merge of shard results.
    python tools/fuzz_topk.py --seconds 120 [--seed S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT]


def make_problem(rng):
    d = int(rng.choice([40, 100, 128, 129, 200, 256, 300, 448, 555, 700]))
    n = int(rng.integers(16384, 26000)) if rng.random() < 0.75 else int(rng.integers(1000, 16384))     # product: fused kernels from 16,384 rows, distance slab below
    q = int(rng.integers(256, 700)) if rng.random() < 0.7 else int(rng.integers(1, 256))
    k = int(rng.choice([1, 10, 64, 100, 251, 300, 512]))
    k = min(k, n)
    metric = int(rng.integers(0, 2))
    kind = rng.integers(0, 4)
    g = rng.standard_normal((n, d)).astype(np.float32)
    if kind == 1:        # clustered: class centres + noise (real feature sets look like this)
        centres = rng.standard_normal((50, d)).astype(np.float32)
        g = (centres[rng.integers(0, 50, size=n)] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    elif kind == 2:      # duplicate-heavy: tie groups straddling rank k, lists that overflow
        g = g[rng.integers(0, max(n // 40, 2), size=n)]
    elif kind == 3:      # a few irregular rows
        for r in rng.integers(0, n, size=5):
            g[r, rng.integers(0, d)] = rng.choice([np.nan, np.inf, -np.inf, 3e30])
    qs = g[rng.permutation(n)[:q]].copy() if rng.random() < 0.6 else rng.standard_normal((q, d)).astype(np.float32)
    kb = None
    if d > 256 and rng.random() < 0.5:
        a = int(rng.integers(1, d // 4)) * 4
        kb = [a, d - a]
    off = int(rng.choice([0, 0, 12345]))
    return qs, g, k, metric, kb, off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    import sehip
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    calls = merges = 0
    while time.time() - t0 < args.seconds:
        qs, g, k, metric, kb, off = make_problem(rng)
        if metric == 0:
            with np.errstate(all="ignore"):
                g = ro.canon_normalize_rows(g); qs = ro.canon_normalize_rows(qs)
        with np.errstate(all="ignore"):
            pd = ro.canon_pdist(qs, g, metric, kblocks=kb)
            wd, wi = ro.canon_topk_rows(pd, k, col_offset=off)
        gq, gg = torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda()
        dd, ii = sehip.retrieve_topk(gq, gg, k, metric=metric, kblocks=kb, col_offset=off)
        ok = np.array_equal(ii.cpu().numpy(), wi) and np.array_equal(dd.cpu().numpy(), wd, equal_nan=True)
        if not ok:
            print("MISMATCH retrieve_topk: q=%d n=%d d=%d k=%d metric=%d kblocks=%s off=%d seed=%d call=%d" % (qs.shape[0], g.shape[0], g.shape[1], k, metric, kb, off, args.seed, calls))
            sys.exit(1)
        calls += 1
        # the same problem as 3 gallery shards + merge
        if g.shape[0] >= 3 * k and rng.random() < 0.5:
            bounds = [0, g.shape[0] // 3, 2 * g.shape[0] // 3, g.shape[0]]
            parts_d, parts_i = [], []
            for s in range(3):
                d_s, i_s = sehip.retrieve_topk(gq, gg[bounds[s]:bounds[s + 1]], k, metric=metric, kblocks=kb, col_offset=off + bounds[s])
                parts_d.append(d_s); parts_i.append(i_s)
            md, mi = sehip.topk_merge(torch.stack(parts_d), torch.stack(parts_i))
            if not (np.array_equal(mi.cpu().numpy(), wi) and np.array_equal(md.cpu().numpy(), wd, equal_nan=True)):
                print("MISMATCH sharded + merge: q=%d n=%d d=%d k=%d metric=%d seed=%d call=%d" % (qs.shape[0], g.shape[0], g.shape[1], k, metric, args.seed, calls))
                sys.exit(1)
            merges += 1
    print("fuzz_topk: %d calls (+ %d three-shard merges), all bit-equal to the oracle (seed %d, %.0f s)" % (calls, merges, args.seed, args.seconds))


if __name__ == "__main__":
    main()
