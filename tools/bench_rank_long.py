"""Rows above 53,248 columns: sorted runs + merge vs the tiled kernel (tuning library: SE_RANK_NORUNS=1 pins the latter).
usage: python tools/bench_rank_long.py [check] [time]"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semantic-embeddings_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")]
import numpy as np, torch, sehip
from oracle import retrieval_oracle as ro

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

what = sys.argv[1:] or ["check", "time"]
if "check" in what:
    for n in (53249, 60000, 65537, 81921, 100353, 106496, 106497, 131072, 150000, 212992, 212993, 300000, 425984):
        rng = np.random.default_rng(n)
        pd = rng.standard_normal((5, n)).astype(np.float32)
        pd[1] = rng.integers(-2, 3, size=n).astype(np.float32)
        pd[2, ::3] = pd[2, 0]
        pd[2, 1:9] = np.array([np.nan, 0.0, -0.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38], dtype=np.float32)
        pd[3] = np.abs(pd[3]); pd[4, n // 2:] = pd[4, :n - n // 2]
        for idx64 in (False, True):
            got = sehip.rank_rows(torch.from_numpy(pd).cuda(), idx64=idx64).cpu().numpy()
            ok = np.array_equal(got, ro.canon_rank_rows(pd))
            print("n=%d idx64=%d %s" % (n, idx64, "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                w = ro.canon_rank_rows(pd)
                for r in range(5):
                    bad = np.nonzero(got[r] != w[r])[0]
                    if len(bad): print("  row", r, "first bad", bad[:5], got[r][bad[:5]], w[r][bad[:5]], "count", len(bad))
    x = torch.randn(1500, 70001, device="cuda")
    x[::2, ::5] = 0.25
    got = sehip.rank_rows(x)
    want = torch.argsort(x, dim=1, stable=True)
    print("1500 x 70001 vs torch stable argsort:", bool((got.long() == want).all()))
if "time" in what:
    for q, n in ((8192, 60000), (8192, 100000), (8192, 106496), (4096, 150000), (4096, 200000), (2048, 400000), (1024, 500000)):
        x = torch.randn(q, n, device="cuda")
        ms = timeit(lambda: sehip.rank_rows(x))
        print("rank_rows %d x %d: %.2f ms = %.2f ps/key  (50k x 50k rate: 3.76 ps/key)" % (q, n, ms, ms * 1e9 / (q * n)), flush=True)
