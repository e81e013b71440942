"""End-to-end timing of the retrieval evaluation the CLI performs (evaluate_retrieval.main -> ClassHierarchy.hierarchical_precision_device):
N synthetic CIFAR-100-like features, the CIFAR hierarchy of tests/golden/hierarchy_cifar.npz, ks = 1..250, whole-list AHP + AP.
Prints wall time per phase (cProfile, top entries) -- where the seconds go once the kernels take milliseconds."""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--per-query", action="store_true")
    ap.add_argument("--clip-ahp", type=int, default=0, help="AHP@K and no AP (the CLI's --clip_ahp K --skip_ap): only the head of every ranking is needed")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--full-ranking", action="store_true", help="with --clip-ahp: rank everything anyway (the path before the fused top-L one)")
    ap.add_argument("--euclid", action="store_true", help="the CLI's default branch (--norm no): Euclidean distances")
    ap.add_argument("--reps", type=int, default=3, help="timed evaluations (the first one after the warm-up allocates the whole-matrix tile cache)")
    args = ap.parse_args()
    import torch
    from class_hierarchy import ClassHierarchy
    g = np.load(os.path.join(ROOT, "tests", "golden", "hierarchy_cifar.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for p, c in g["edges"]:
            f.write("%d %d\n" % (p, c))
    h = ClassHierarchy.from_file(f.name, id_type=int)
    os.unlink(f.name)
    rng = np.random.default_rng(0)
    classes = sorted(set(g["labels"].tolist()))
    labels = [classes[i] for i in rng.integers(0, len(classes), size=args.n)]
    centers = rng.standard_normal((max(classes) + 1, args.d)).astype(np.float32)
    feats = (centers[labels] + 0.8 * rng.standard_normal((args.n, args.d))).astype(np.float32)
    ks = list(range(1, 251))
    kw = dict(compute_ahp=args.clip_ahp or True, compute_ap=not args.clip_ahp, normalize=not args.euclid)
    if not args.per_query:
        kw["per_query"] = False
    if args.full_ranking:
        kw["head_via_topk"] = False
    h.hierarchical_precision_device(feats[:4096].copy(), labels[:4096], ks, **kw)        # warm-up (library load, allocator)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if args.profile else None
    for rep in range(args.reps):
        t0 = time.perf_counter()
        if pr and rep == args.reps - 1:
            pr.enable()
        avg, _ = h.hierarchical_precision_device(feats.copy(), labels, ks, **kw)
        torch.cuda.synchronize()
        if pr and rep == args.reps - 1:
            pr.disable()
        dt = time.perf_counter() - t0
        print("n=%d d=%d %s per_query=%s clip_ahp=%d call %d: %.3f s end to end   %s" %
              (args.n, args.d, "euclid" if args.euclid else "cosine", args.per_query, args.clip_ahp, rep + 1, dt,
               "  ".join("%s %.6f" % (k, v) for k, v in avg.items() if not k.startswith("P@") or k.startswith("P@1 "))))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
