#!/usr/bin/env python
"""Kernel-level microbenchmarks (HIP-event timing on the launch stream) used while tuning.
    python tools/bench_kernels.py pdist|rank|loss|topk [--n 50000 --d 100 --reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd"))
import sehip  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["pdist", "rank", "loss", "topk", "fused", "hprec", "shard", "rownorm"])
    ap.add_argument("--hp-mode", default="all", choices=["all", "whole", "sweep"], help="hprec: every configuration, or whole-list AHP + AP in class order only (profiling)")
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--q", type=int, default=None)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--k", type=int, default=251)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n, d = args.n, args.d
    q = args.q or n
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)).cuda()
    sehip.normalize_rows_(x)
    if args.what == "rownorm":
        # se_normalize_rows / se_row_sqnorm at the retrieval shapes: CIFAR (50k x 100: one lane per row) and one ILSVRC gallery
        # shard (160,146 x 1000: one wave per row)
        for rows, dd in ((50000, 100), (160146, 1000), (160146, 555), (40000, 4096)):
            y = torch.from_numpy(np.random.default_rng(3).standard_normal((rows, dd)).astype(np.float32)).cuda()
            med, mn = timeit(lambda: sehip.normalize_rows_(y), args.reps)
            print("normalize_rows %d x %d: median %.3f ms (min %.3f)  %.2f TB/s (read + write)" % (rows, dd, med, mn, 8.0 * rows * dd / med / 1e9))
            med, mn = timeit(lambda: sehip.row_sqnorm(y), args.reps)
            print("row_sqnorm     %d x %d: median %.3f ms (min %.3f)  %.2f TB/s (read)" % (rows, dd, med, mn, 4.0 * rows * dd / med / 1e9))
        return
    if args.what == "pdist":
        out = torch.empty((q, n), dtype=torch.float32, device="cuda")
        y = x.clone()
        for name, b in (("symmetric(a==b)", x), ("general(a!=b)", y)):
            med, mn = timeit(lambda: sehip.pairwise_dist(x[:q], b, metric=sehip.METRIC_COSINE, out=out), args.reps)
            gb = (4.0 * q * n + 4.0 * (q + n) * d) / 1e9
            print("pdist %-16s q=%d n=%d d=%d: median %.3f ms (min %.3f)  %.1f GB/s algorithmic, %.1f TFLOP/s useful" %
                  (name, q, n, d, med, mn, gb / med * 1e3, 2.0 * q * n * d / med / 1e9))
        xe = torch.from_numpy(np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)).cuda()
        sq = sehip.row_sqnorm(xe)
        for name, b in (("Euclid symmetric", xe), ("Euclid general", xe.clone())):
            med, mn = timeit(lambda: sehip.pairwise_dist(xe[:q], b, metric=sehip.METRIC_EUCLID, sqa=sq[:q], sqb=sq, out=out), args.reps)
            print("pdist %-16s q=%d n=%d d=%d: median %.3f ms (min %.3f)  %.1f GB/s algorithmic" % (name, q, n, d, med, mn, gb / med * 1e3))
    elif args.what == "rank":
        pd = sehip.pairwise_dist(x[:q], x, metric=sehip.METRIC_COSINE)
        rk = torch.empty((q, n), dtype=torch.int32, device="cuda")
        med, mn = timeit(lambda: sehip.rank_rows(pd, out=rk), args.reps)
        print("rank q=%d n=%d: median %.3f ms (min %.3f)  %.1f GB/s algorithmic, %.1f Mkeys/s" % (q, n, med, mn, 8.0 * q * n / med / 1e6, q * n / med / 1e3))
        # the CLI-default Euclidean branch: all-positive distances (skewed top digit -> the group-peeling build)
        xe = torch.from_numpy(np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)).cuda()
        sq = sehip.row_sqnorm(xe)
        pd = sehip.pairwise_dist(xe[:q], xe, metric=sehip.METRIC_EUCLID, sqa=sq[:q], sqb=sq, out=pd)
        med, mn = timeit(lambda: sehip.rank_rows(pd, out=rk), args.reps)
        print("rank (Euclidean rows) q=%d n=%d: median %.3f ms (min %.3f)  %.1f GB/s algorithmic" % (q, n, med, mn, 8.0 * q * n / med / 1e6))
        # rows above 53,248 columns: two sorted runs + merge (real cosine rows: 8,192 queries against a 100,000-row gallery)
        del pd, rk
        xl = torch.from_numpy(np.random.default_rng(2).standard_normal((100000, d)).astype(np.float32)).cuda()
        sehip.normalize_rows_(xl)
        pdl = sehip.pairwise_dist(xl[:8192], xl, metric=sehip.METRIC_COSINE)
        rkl = torch.empty((8192, 100000), dtype=torch.int32, device="cuda")
        med, mn = timeit(lambda: sehip.rank_rows(pdl, out=rkl), args.reps)
        print("rank (long rows) q=8192 n=100000: median %.3f ms (min %.3f)  %.2f ps per key, %.1f GB/s algorithmic" % (med, mn, med * 1e9 / (8192 * 100000), 8.0 * 8192 * 100000 / med / 1e6))
    elif args.what == "hprec":
        C = 100
        rng = np.random.default_rng(1)
        cls = torch.from_numpy(rng.integers(0, C, size=n).astype(np.int32)).cuda()
        tab = rng.random((C, C)); tab = (tab + tab.T) / 2; np.fill_diagonal(tab, 1.0)
        counts = np.bincount(cls.cpu().numpy(), minlength=C)
        best = np.stack([np.cumsum(np.repeat(tab[c][np.argsort(-tab[c], kind="stable")], counts[np.argsort(-tab[c], kind="stable")])) for c in range(C)])
        tab_d, best_d = torch.from_numpy(tab).cuda(), torch.from_numpy(best).cuda()
        qq = min(q, 8192) if args.q is None else q          # (--q given: that many queries, e.g. the full 50,000)
        pd = sehip.pairwise_dist(x[:qq], x, metric=sehip.METRIC_COSINE)
        rk = sehip.rank_rows(pd)
        ks = torch.arange(1, 251, dtype=torch.int32, device="cuda")
        qidx = torch.arange(qq, dtype=torch.int32, device="cuda")
        curves = sehip.hprec_reciprocal_curves(best_d, best_d)
        qcls = cls[:qq].contiguous()
        for name, ahp in (("whole-list AHP + AP", 0), ("AHP@250, no AP", 250))[:1 if args.hp_mode != "all" else 2]:
            for order in (True, False)[:1 if args.hp_mode != "all" else 2]:
                med, mn = timeit(lambda: sehip.hierarchical_precision(rk, cls, qcls, qidx, tab_d, tab_d, best_d, best_d, ks, ahp_len=ahp,
                                                                      want_ap=(ahp == 0), curves=curves, class_order=order), args.reps)
                print("hprec %-22s %-14s q=%d n=%d: median %.3f ms (min %.3f)  %.1f GB/s of ranks, %.1f Mranks/s" %
                      (name, "class order" if order else "query order", qq, n, med, mn, 4.0 * qq * n / med / 1e6, qq * n / med / 1e3))
        if args.hp_mode == "sweep":     # per-query overhead (intercept) vs per-rank cost (slope)
            for ll in (512, 4096, 8192, 16384, 32768, n):
                med, mn = timeit(lambda: sehip.hierarchical_precision(rk, cls, qcls, qidx, tab_d, tab_d, best_d, best_d, ks, ahp_len=0, want_ap=True,
                                                                      curves=curves, class_order=True, list_len=ll), args.reps)
                print("hprec sweep list_len=%d: median %.3f ms" % (ll, med))
        med, mn = timeit(lambda: sehip.hprec_reciprocal_curves(best_d, best_d), args.reps)
        print("hprec reciprocal curves C=%d n=%d: median %.3f ms" % (C, n, med))
    elif args.what == "shard":
        # BASELINE.json configs[4], one rank's share: 50,000 queries x (1,281,167 / 8) gallery rows, D = 1000, top-251, then the
        # merge of the 8 all-gathered lists (synthetic: 8 copies with shifted indices)
        D, Q, NS, K = 1000, args.q or 50000, 160146, args.k
        g = torch.from_numpy(np.random.default_rng(1).standard_normal((NS, D)).astype(np.float32)).cuda()
        qq = torch.from_numpy(np.random.default_rng(2).standard_normal((Q, D)).astype(np.float32)).cuda()
        sehip.normalize_rows_(g); sehip.normalize_rows_(qq)
        med, mn = timeit(lambda: sehip.retrieve_topk(qq, g, K, metric=sehip.METRIC_COSINE, col_offset=NS), max(2, args.reps // 2))
        print("shard retrieve_topk (one chain) q=%d n=%d d=%d k=%d: median %.1f ms (min %.1f)  %.1f Mpairs/s, %.1f TFLOP/s useful" %
              (Q, NS, D, K, med, mn, Q * NS / med / 1e3, 2.0 * Q * NS * D / med / 1e9))
        kb = [448, 276, 276]                  # evaluate_retrieval.host_blas_kblocks(1000): the arithmetic that reproduces np.dot at this depth
        med, mn = timeit(lambda: sehip.retrieve_topk(qq, g, K, metric=sehip.METRIC_COSINE, col_offset=NS, kblocks=kb), max(2, args.reps // 2))
        print("shard retrieve_topk (K-blocks %s) q=%d n=%d d=%d k=%d: median %.1f ms (min %.1f)  %.1f Mpairs/s, %.1f TFLOP/s useful" %
              (kb, Q, NS, D, K, med, mn, Q * NS / med / 1e3, 2.0 * Q * NS * D / med / 1e9))
        od, oi = sehip.retrieve_topk(qq, g, K, metric=sehip.METRIC_COSINE)
        dd = torch.stack([od + 1e-3 * r for r in range(8)]); ii = torch.stack([oi + NS * r for r in range(8)])
        med, mn = timeit(lambda: sehip.topk_merge(dd, ii), args.reps)
        print("topk_merge parts=8 q=%d k=%d: median %.3f ms  (all-gather payload %.1f MB per rank)" % (Q, K, med, Q * K * 8 / 1e6))
    elif args.what == "fused":
        # se_retrieve_topk, all-pairs (queries == gallery): distances + top-k with no [q, n] matrix
        for metric, name in ((sehip.METRIC_COSINE, "cosine"), (sehip.METRIC_EUCLID, "Euclid")):
            sq = sehip.row_sqnorm(x) if metric == sehip.METRIC_EUCLID else None
            med, mn = timeit(lambda: sehip.retrieve_topk(x[:q], x, args.k, metric=metric, sqq=None if sq is None else sq[:q], sqg=sq), args.reps)
            print("fused retrieve_topk %-6s q=%d n=%d d=%d k=%d: median %.3f ms (min %.3f)  %.1f Mpairs/s, %.1f TFLOP/s (fp32 MFMA), %.2f GB algorithmic" %
                  (name, q, n, d, args.k, med, mn, q * n / med / 1e3, 2.0 * q * n * d / med / 1e9, (4.0 * (q + n) * d + 8.0 * q * args.k) / 1e9))
    elif args.what == "topk":
        pd = sehip.pairwise_dist(x[:q], x, metric=sehip.METRIC_COSINE)
        med, mn = timeit(lambda: sehip.topk_rows(pd, args.k), args.reps)
        print("topk k=%d q=%d n=%d: median %.3f ms (min %.3f)  %.1f GB/s" % (args.k, q, n, med, mn, 4.0 * q * n / med / 1e6))
    else:
        B, D = 65536, 1000
        xx = torch.randn(B, D, device="cuda")
        E = torch.nn.functional.normalize(torch.randn(1000, D, device="cuda"), dim=-1)
        yy = torch.randint(0, 1000, (B,), device="cuda")
        med, mn = timeit(lambda: sehip.cosine_loss_forward(xx, yy, E), args.reps)
        print("loss fwd B=%d D=%d f32: median %.3f ms  %.1f GB/s" % (B, D, med, (B * D * 12.0) / med / 1e6))
        med, mn = timeit(lambda: sehip.cosine_loss_backward(xx, yy, E, grad_scale=1.0 / B), args.reps)
        print("loss bwd B=%d D=%d f32: median %.3f ms  %.1f GB/s" % (B, D, med, (B * D * 12.0) / med / 1e6))
        xb = xx.bfloat16()
        med, mn = timeit(lambda: sehip.cosine_loss_forward(xb, yy, E, want_xhat=False), args.reps)
        print("loss fwd (no xhat) bf16: median %.3f ms  %.1f GB/s" % (med, (B * D * 6.0) / med / 1e6))
        # DeViSE ranking loss (utils.py:103-122), forward + backward: the fused MFMA kernels against the same loss written in torch ops
        for C2, B3 in ((100, 128), (1000, 128), (1000, 1024)):
            Ed = torch.nn.functional.normalize(torch.randn(C2, C2, device="cuda"), dim=-1)
            yp = torch.nn.functional.normalize(torch.randn(B3, C2, device="cuda"), dim=-1).requires_grad_(True)
            yl = torch.randint(0, C2, (B3,), device="cuda")

            def hip_step():
                yp.grad = None
                sehip.devise_ranking_loss(yp, yl, Ed, 0.1).mean().backward()

            def torch_step():
                yp.grad = None
                true = (yp * Ed[yl]).sum(-1)
                (torch.relu(0.1 - true[:, None] + yp @ Ed.t()).sum(-1) - 0.1).mean().backward()
            mh, _ = timeit(hip_step, 20)
            mt, _ = timeit(torch_step, 20)
            print("devise fwd+bwd B=%d C=D=%d: HIP %.1f us, torch ops %.1f us" % (B3, C2, mh * 1e3, mt * 1e3))
        B2 = 128
        x2, y2, E2 = torch.randn(B2, 100, device="cuda"), torch.randint(0, 100, (B2,), device="cuda"), E[:100, :100].contiguous()
        med, mn = timeit(lambda: sehip.cosine_loss_forward(x2, y2, E2), 20)
        print("loss fwd B=128 D=100: median %.1f us" % (med * 1e3))


if __name__ == "__main__":
    main()
