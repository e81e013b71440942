#!/usr/bin/env python
"""Development driver for the image path of se_rank_rows (VAR 3): crafted rows against the canonical oracle, and timing.
    SEHIP_LIB=.../libsehip_tuning.so [SE_RANK_PEEL=3] python tools/dev_img.py check|time [--n 50000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT]


from tests.test_gpu_retrieval import image_path_rows as crafted_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["check", "time"])
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--q", type=int, default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rows", default="cosine", choices=["cosine", "uniform"], help="time: -dot of normalised gaussian features, or uniform(-1, 1) keys")
    args = ap.parse_args()
    import torch
    import sehip
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(args.seed)
    if args.what == "check":
        bad_total = 0
        for n in (args.n, 36000, 33001, 45000, 50176, 40961):
            pd = crafted_rows(rng, n)
            # more rows than workgroups would make no difference here; repeat the block so that a workgroup sees several rows (back-off state)
            pd = np.concatenate([pd, pd[::-1]], axis=0)
            t = time.time()
            got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()
            want = ro.canon_rank_rows(pd)
            bad = [r for r in range(pd.shape[0]) if not np.array_equal(got[r], want[r])]
            for r in bad[:4]:
                d = np.nonzero(got[r] != want[r])[0]
                print("  n=%d row %d: %d positions differ, first at %d: got %s want %s" % (n, r, d.size, d[0], got[r][d[0]:d[0] + 4], want[r][d[0]:d[0] + 4]))
            print("n=%d: %d rows, %d differ (%.1f s)" % (n, pd.shape[0], len(bad), time.time() - t))
            bad_total += len(bad)
        # many rows per workgroup: 600 cosine rows with a few unfit rows mixed in
        n = args.n
        x = rng.standard_normal((n, 64)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        pd = -(x[:600] @ x.T)
        pd[5::97] = rng.choice(np.array([0.5, -0.25], dtype=np.float32), size=(len(pd[5::97]), n))
        got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()
        want = ro.canon_rank_rows(pd)
        bad = [r for r in range(pd.shape[0]) if not np.array_equal(got[r], want[r])]
        print("600 cosine rows (+ unfit ones): %d differ" % len(bad))
        bad_total += len(bad)
        print("CHECK %s" % ("OK" if bad_total == 0 else "FAILED"))
        sys.exit(1 if bad_total else 0)
    n = args.n
    q = args.q or n
    x = torch.from_numpy(rng.standard_normal((n, 100)).astype(np.float32)).cuda()
    sehip.normalize_rows_(x)
    pd = sehip.pairwise_dist(x[:q], x, metric=sehip.METRIC_COSINE)
    if args.rows == "uniform":
        pd.uniform_(-1.0, 1.0)
    rk = torch.empty((q, n), dtype=torch.int32, device="cuda")

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts)), float(np.min(ts))
    med, mn = timeit(lambda: sehip.rank_rows(pd, out=rk))
    print("rank cosine q=%d n=%d: median %.3f ms (min %.3f)" % (q, n, med, mn))
    print("order guard violations: %d" % sehip.rank_rows_check(pd, rk))
    samp = np.linspace(0, q - 1, 24).astype(int)
    want = ro.canon_rank_rows(pd[samp].cpu().numpy())
    print("sampled rows equal to the oracle: %s" % np.array_equal(rk[samp].cpu().numpy(), want))
    xe = torch.from_numpy(rng.standard_normal((n, 100)).astype(np.float32)).cuda()
    sq = sehip.row_sqnorm(xe)
    pd = sehip.pairwise_dist(xe[:q], xe, metric=sehip.METRIC_EUCLID, sqa=sq[:q], sqb=sq, out=pd)
    med, mn = timeit(lambda: sehip.rank_rows(pd, out=rk))
    print("rank Euclid q=%d n=%d: median %.3f ms (min %.3f)" % (q, n, med, mn))
    print("order guard violations: %d" % sehip.rank_rows_check(pd, rk))


if __name__ == "__main__":
    main()
