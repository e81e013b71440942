"""Randomised check of se_hierarchical_precision against the NumPy statement of the reference's per-query loop
(tests/test_dp_gloo.py:_hprec_standin, class_hierarchy.py:257-314): list lengths across the chunk size, class counts for all three
class-table modes, random / consecutive / duplicate cut-offs, whole-list and clipped AHP, AP, queries anywhere in (or absent from)
their rankings, both visiting orders.
    python tools/fuzz_hprec.py --seconds 120 [--seed S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "semantic-embeddings_amd"), ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    import sehip
    from test_dp_gloo import _hprec_standin
    rng = np.random.default_rng(args.seed)
    t0, cases, worst = time.time(), 0, 0.0
    while time.time() - t0 < args.seconds:
        n = int(rng.choice([rng.integers(2, 300), rng.integers(300, 5000), 4095, 4096, 4097, 8192, rng.integers(8000, 20000)]))
        q = int(min(n, rng.integers(1, 40)))
        C = int(rng.choice([2, 7, 100, 256, 257, 900]))
        cls = rng.integers(0, C, size=n).astype(np.int32)
        tw = rng.random((C, C)) * 0.8 + 0.1; tw = (tw + tw.T) / 2; np.fill_diagonal(tw, 1.0)
        tl = rng.random((C, C)) * 0.8 + 0.1; tl = (tl + tl.T) / 2; np.fill_diagonal(tl, 1.0)
        counts = np.bincount(cls, minlength=C)

        def best(tab):
            out = np.empty((C, n))
            for c in range(C):
                o = np.argsort(-tab[c], kind="stable")
                out[c] = np.cumsum(np.repeat(tab[c][o], counts[o]))
            return out
        bw, bl = best(tw), best(tl)
        rk = np.stack([rng.permutation(n) for _ in range(q)]).astype(np.int32)
        qidx = np.arange(q, dtype=np.int32)
        mode = rng.integers(0, 3)
        for r in range(q):
            if mode == 0 or (mode == 2 and r % 2 == 0):      # the query is its own nearest neighbour
                at = np.flatnonzero(rk[r] == r)[0]
                rk[r, at], rk[r, 0] = rk[r, 0], r
            if rng.random() < 0.1:
                qidx[r] = -1                                  # not part of the gallery
        eff = n - 1                                          # shortest effective list
        if eff < 1:
            continue
        nk = int(rng.integers(0, 9))
        style = rng.integers(0, 3)
        if style == 0:
            ks = np.arange(1, min(eff, int(rng.integers(1, 400))) + 1)
        elif style == 1:
            ks = rng.integers(1, eff + 1, size=nk)
        else:
            ks = np.sort(rng.integers(1, eff + 1, size=nk))
        ks = ks.astype(np.int32)[:512]
        ahp = int(rng.choice([-1, 0, 0, min(eff, 50), int(rng.integers(1, eff + 1)), eff + 100]))
        ap_ = bool(rng.integers(0, 2))
        order = [None, True, False][int(rng.integers(0, 3))]
        args_h = [torch.from_numpy(a) for a in (rk, cls, cls[:q].copy(), qidx, tw, tl, bw, bl, ks)]
        want = _hprec_standin(*args_h, ahp_len=ahp, want_ap=ap_).numpy()
        dev = [a.cuda() for a in args_h]
        got = sehip.hierarchical_precision(*dev, ahp_len=ahp, want_ap=ap_, class_order=order).cpu().numpy()
        err = float(np.abs(got - want).max()) if got.size else 0.0
        worst = max(worst, err)
        if not err <= 1e-10:
            np.savez("gpurun_out/fuzz_hprec_fail.npz", rk=rk, cls=cls, qidx=qidx, tw=tw, tl=tl, ks=ks, ahp=ahp, ap=ap_)
            print("MISMATCH case %d: n=%d q=%d C=%d nk=%d ahp=%d ap=%s order=%s err=%.3e" % (cases, n, q, C, len(ks), ahp, ap_, order, err))
            sys.exit(1)
        cases += 1
    print("fuzz_hprec: %d calls, worst |difference| %.2e (bound 1e-10; seed %d, %.0f s)" % (cases, worst, args.seed, time.time() - t0))


if __name__ == "__main__":
    main()
