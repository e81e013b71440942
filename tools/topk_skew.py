#!/usr/bin/env python
"""se_retrieve_topk on a CLASS-SORTED clustered gallery (a query's neighbours sit in a few adjacent gallery tiles) against the same
gallery shuffled: time, queries redone exactly, candidates per query (the library's own counters, se_phase_timing_read).
    python tools/topk_skew.py [--n 50000 --d 100 --classes 100 --k 251]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--q", type=int, default=None)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--k", type=int, default=251)
    ap.add_argument("--noise", type=float, default=0.05)
    args = ap.parse_args()
    import torch
    import sehip
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(0)
    n, d, C = args.n, args.d, args.classes
    cen = rng.standard_normal((C, d)).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    y = np.sort(rng.integers(0, C, size=n))
    x = (cen[y] + args.noise * rng.standard_normal((n, d))).astype(np.float32)
    perm = rng.permutation(n)
    for name, feats in (("class-sorted", x), ("shuffled", x[perm])):
        g = torch.from_numpy(feats).cuda()
        sehip.normalize_rows_(g)
        qs = g if args.q is None else g[: args.q].clone()
        run = lambda: sehip.retrieve_topk(qs, g, args.k)   # noqa: E731
        dd, ii = run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        sehip.phase_timing(True)
        run()
        ph, cnt = sehip.phase_timing_read()
        sehip.phase_timing(False)
        rows = np.linspace(0, qs.shape[0] - 1, 8).astype(int)
        gh = g.cpu().numpy()
        pd = -(gh[rows].astype(np.float32) @ gh.T)          # not the canonical chain: only the head's SET is compared loosely below
        ok = all(len(set(ii[r].cpu().numpy().tolist()) & set(np.argsort(pd[j], kind="stable")[: args.k].tolist())) >= args.k - 8 for j, r in enumerate(rows))
        print("%-13s q=%d n=%d d=%d k=%d: %.2f ms; redone exactly %s of %s queries, %.0f candidates / %.0f recomputed per query; phases %s; heads plausible: %s"
              % (name, qs.shape[0], n, d, args.k, float(np.median(ts)), cnt["redone"], cnt["queries"], cnt["candidates"] / max(1, cnt["queries"]),
                 cnt["recomputed"] / max(1, cnt["queries"]), {k: round(v, 2) for k, v in ph.items()}, ok))


if __name__ == "__main__":
    main()
