"""Randomised check of se_rank_rows against the canonical oracle (oracle/canon.c), aimed at the variants of the register-resident kernel:
row lengths around every instantiation boundary, value mixes that send a call to the plain / group-peeling / two-pass variant, and rows
inside such a call that do not fit it (keys below the two-pass window: 0 .. 300 of them, ties, zeros, negatives, NaN, infinities).
    python tools/fuzz_rank.py --seconds 120 [--seed S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT]


def make_rows(rng, q, n):
    kind = rng.integers(0, 9)
    if kind == 0:      # Euclidean-like: positive, two or three exponents
        base = rng.choice([3.0, 50.0, 200.0, 1e4, 1e-3])
        pd = (base * (1.0 + 0.12 * rng.standard_normal((q, n)))).astype(np.float32)
        pd = np.abs(pd) + np.float32(base * 0.3)
    elif kind == 1:    # cosine-like: -dot, both signs
        pd = (0.1 * rng.standard_normal((q, n))).astype(np.float32)
    elif kind == 2:    # few distinct values
        pd = rng.choice((100.0 + rng.integers(0, 40, size=9)).astype(np.float32), size=(q, n))
    elif kind == 3:    # negative narrow range (window below a negative maximum)
        pd = (-200.0 + 20.0 * rng.standard_normal((q, n))).astype(np.float32)
    elif kind == 4:    # wide positive
        pd = np.exp(rng.uniform(-20, 20, size=(q, n))).astype(np.float32)
    elif kind == 5:    # image path: a 2^-24 grid around zero under one key of magnitude 1 -- colliding images at a density the draw picks
        span = 1 << int(rng.integers(14, 24))
        pd = ((rng.integers(0, span, size=(q, n)) - span // 2).astype(np.float32) * np.float32(2.0 ** -24))
        pd[:, int(rng.integers(0, n))] = rng.choice(np.array([1.0, -1.0, 0.7, -1.0000001], dtype=np.float32))
    elif kind == 6:    # image path: clustered (a few centres + tiny noise: long runs of near-ties inside few images)
        cen = (0.2 * rng.standard_normal(int(rng.integers(5, 200)))).astype(np.float32)
        pd = (rng.choice(cen, size=(q, n)) + np.float32(10.0 ** rng.uniform(-9, -5)) * rng.standard_normal((q, n))).astype(np.float32)
    elif kind == 7:    # image path: cosine-like with every key repeated 2 .. 20 times (exact-tie runs, index order)
        rep = int(rng.integers(2, 21))
        pd = np.repeat((0.1 * rng.standard_normal((q, n // rep + 1))).astype(np.float32), rep, axis=1)[:, :n]
        pd = np.take_along_axis(pd, rng.permuted(np.tile(np.arange(n), (q, 1)), axis=1), axis=1)
    else:              # image path: all keys of a row inside one binade of either sign
        pd = (rng.choice(np.array([1.0, -1.0], dtype=np.float32), size=(q, 1)) * (1.0 + rng.random((q, n)))).astype(np.float32) * np.float32(2.0 ** int(rng.integers(-20, 20)))
    for r in range(q):
        m = int(rng.choice([0, 0, 1, 2, 17, 255, 256, 257, 300]))
        if m:
            cols = rng.choice(n, size=m, replace=False)
            low = rng.choice(np.array([0.0, -0.0, 1e-30, -1e-30, -5.0, -np.inf, 1e-3], dtype=np.float32), size=m)
            if rng.random() < 0.5:
                low = (rng.standard_normal(m) * 1e-2).astype(np.float32)
            pd[r, cols] = low
        if rng.random() < 0.2:
            pd[r, rng.choice(n, size=int(rng.integers(1, 5)), replace=False)] = rng.choice(
                np.concatenate([np.array([np.nan, np.inf], dtype=np.float32), np.array([0xFFC00000, 0xFF800001], dtype=np.uint32).view(np.float32)]))   # NaNs of both signs
        if rng.random() < 0.2:
            pd[r, ::int(rng.integers(2, 9))] = pd[r, 0]
    return np.ascontiguousarray(pd, dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--long", action="store_true", help="rows above 53,248 columns: sorted runs + merge tree (2 / 4 / 8 segments)")
    args = ap.parse_args()
    import torch
    import sehip
    from oracle import retrieval_oracle as ro
    rng = np.random.default_rng(args.seed)
    t0, cases, rows = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        n = int(rng.choice([rng.integers(32768, 53249), 32768, 32769, 40960, 40961, 50000, 53248, rng.integers(700, 32768)]))
        if args.long:   # every segment count, segment lengths around the instantiation boundaries of the segment kernel
            n = int(rng.choice([rng.integers(53249, 106497), rng.integers(106497, 212993), rng.integers(212993, 425985), 53249, 65536, 65537,
                                81920, 81921, 100352, 100353, 106496, 106497, 131072, 212992, 212993, 425984]))
        q = int(rng.integers(1, 9))
        pd = make_rows(rng, q, n)
        got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()
        want = ro.canon_rank_rows(pd)
        if not np.array_equal(got, want):
            bad = [r for r in range(q) if not np.array_equal(got[r], want[r])]
            np.save("gpurun_out/fuzz_rank_fail.npy", pd)
            print("MISMATCH case %d: n=%d q=%d rows %s (input saved to gpurun_out/fuzz_rank_fail.npy)" % (cases, n, q, bad))
            sys.exit(1)
        cases += 1
        rows += q
    print("fuzz_rank: %d calls, %d rows, all bit-equal to the oracle (seed %d, %.0f s)" % (cases, rows, args.seed, time.time() - t0))


if __name__ == "__main__":
    main()
