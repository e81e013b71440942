#!/usr/bin/env python
"""bench.py -- headline benchmark of the cosine-embedding training + retrieval hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload retrieval|train]

One JSON line on stdout (rank 0).  See DESIGN.md section 6 for what a "step" is:

* retrieval (default; BASELINE.json configs[2], "CIFAR-100 50k-query x 50k-gallery retrieval"):
  one step = the whole ``pairwise_retrieval`` device path on synthetic float32 features already
  resident in HBM -- row normalisation, all-pairs cosine distance matrix, full canonical ranking.
  value = query x gallery pairs ranked per second (Mpairs/s), summed over ranks.  With N > 1 every
  rank evaluates its own 50k x 50k feature set (weak scaling, no data-path collective: retrieval jobs
  and the rows of a ranking are independent -- SURVEY.md section 8e row 2).
* train (BASELINE.json configs[1]): ResNet-110-fc cosine-embedding training step, images/s
  (reported in the ``train`` object of the same line when --with-train is given, or as the primary
  metric with --workload train).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "semantic-embeddings_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="retrieval", choices=["retrieval", "train"])
    ap.add_argument("--n", type=int, default=50000, help="gallery rows (retrieval)")
    ap.add_argument("--q", type=int, default=None, help="query rows per rank (retrieval; default = n)")
    ap.add_argument("--d", type=int, default=100, help="feature dimension (retrieval)")
    ap.add_argument("--metric", default="cosine", choices=["cosine", "euclid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=3072)
    ap.add_argument("--with-train", dest="with_train", action="store_true", default=True,
                    help="also time the ResNet-110-fc training step (adds a 'train' object; default on)")
    ap.add_argument("--no-train", dest="with_train", action="store_false")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (train)")
    ap.add_argument("--arch", default="resnet-110-fc")
    return ap.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm
    return rank, world, local


def barrier_sync(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are enqueued on (torch's current
    stream, which is the stream handed to the C ABI)."""

    def __init__(self):
        self.records = {}

    def time(self, name, fn):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.records.setdefault(name, []).append((a, b))
        return out

    def avg_ms(self):
        torch.cuda.synchronize()
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.records.items()}


# ---------------------------------------------------------------------------------------------
# retrieval workload
# ---------------------------------------------------------------------------------------------

def bench_retrieval(args, rank, world):
    import sehip
    n, d = args.n, args.d
    q = args.q or n
    metric = sehip.METRIC_COSINE if args.metric == "cosine" else sehip.METRIC_EUCLID
    # SURVEY.md 8d synthetic features.  Weak scaling: every rank evaluates its OWN feature file of the same shape
    # (rank r: seed r), i.e. N independent `evaluate_retrieval` jobs -- identical work per rank, no data-path collective.
    rng = np.random.default_rng(rank)
    feats_h = rng.standard_normal((n, d)).astype(np.float32)
    gallery0 = torch.from_numpy(feats_h).cuda()
    if q == n:
        queries0 = None        # all-pairs within the feature set (the reference's only mode): symmetric kernel
    else:
        qrng = np.random.default_rng(100 + rank)
        queries0 = torch.from_numpy(qrng.standard_normal((q, d)).astype(np.float32)).cuda()

    pd = torch.empty((q, n), dtype=torch.float32, device="cuda")
    rk = torch.empty((q, n), dtype=torch.int32, device="cuda")
    timer = None

    def step():
        g = gallery0.clone()
        qs = g if queries0 is None else queries0.clone()
        t = timer.time if timer is not None else (lambda name, fn: fn())
        if metric == sehip.METRIC_COSINE:
            t("normalize_rows", lambda: sehip.normalize_rows_(g))
            if qs is not g:
                sehip.normalize_rows_(qs)
            t("pairwise_dist", lambda: sehip.pairwise_dist(qs, g, metric=metric, out=pd))
        else:
            sq = t("row_sqnorm", lambda: sehip.row_sqnorm(g))
            sqq = sq if qs is g else sehip.row_sqnorm(qs)
            t("pairwise_dist", lambda: sehip.pairwise_dist(qs, g, metric=metric, sqa=sqq, sqb=sq, out=pd))
        t("rank_rows", lambda: sehip.rank_rows(pd, out=rk))

    for _ in range(args.warmup):
        step()
    timer = KernelTimer()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync(world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kms = timer.avg_ms()

    pairs_per_step = float(q) * n * world
    value = pairs_per_step * args.steps / elapsed / 1e6
    # algorithmic bytes / flops per launch
    pd_bytes = 4.0 * q * n + 4.0 * (q + n) * d
    pd_flops = 2.0 * q * n * d
    rk_bytes = 4.0 * q * n + 4.0 * q * n
    kernels = {
        "pairwise_dist": {"ms": kms.get("pairwise_dist"), "algorithmic_GB": pd_bytes / 1e9,
                          "GBps": pd_bytes / 1e6 / kms["pairwise_dist"], "frac_hbm": pd_bytes / 1e6 / kms["pairwise_dist"] / HBM_PEAK_GBS,
                          "TFLOPs": pd_flops / 1e9 / kms["pairwise_dist"], "frac_mfma_f32": pd_flops / 1e9 / kms["pairwise_dist"] / MFMA_F32_PEAK_TFLOPS},
        "rank_rows": {"ms": kms.get("rank_rows"), "algorithmic_GB": rk_bytes / 1e9,
                      "GBps": rk_bytes / 1e6 / kms["rank_rows"], "frac_hbm": rk_bytes / 1e6 / kms["rank_rows"] / HBM_PEAK_GBS},
    }
    dominant = max(("pairwise_dist", "rank_rows"), key=lambda k: kms[k])
    tr = pmc_traffic_gb(dominant, q, n, d)
    roofline = {"kernel": dominant, "bound": "hbm", "achieved": kernels[dominant]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dominant]["frac_hbm"],
                "traffic": None if tr is None else tr["bytes"],          # HBM bytes per launch (PMC), vs algorithmic_GB below
                "algorithmic_bytes": kernels[dominant]["algorithmic_GB"] * 1e9, "traffic_detail": tr}
    out = {
        "metric": "retrieval_Mpairs_per_sec", "value": value, "unit": "Mpairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CIFAR-100-sized retrieval: %d queries/GPU x %d gallery, D=%d, %s, "
                               "normalise + all-pairs distance + full canonical ranking" % (q, n, d, args.metric),
                   "queries_per_gpu": q, "gallery": n, "dim": d,
                   "parallelism": "%d independent feature sets, one per GPU (no collective)" % world},
        "roofline": roofline, "kernels": kernels,
    }
    return out, feats_h


def pmc_traffic_gb(kernel, q, n, d):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE with the gfx950 correction of MI355X_MICROARCH.md where it applies, plus WRITE_SIZE, both in KB), or None when
    no profile of this exact shape is committed.  bench.py cannot collect counters itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)[kernel]
        if [rec["q"], rec["n"], rec["d"]] != [q, n, d]:
            return None
        ff = rec.get("fetch_factor", 2)   # gfx950: FETCH_SIZE halves wide coalesced reads (x2); other widths count in full
        return {"bytes": (rec["fetch_kb"] * ff + rec["write_kb"]) * 1024.0, "read_GB": rec["fetch_kb"] * ff * 1024 / 1e9,
                "write_GB": rec["write_kb"] * 1024 / 1e9, "source": rec["source"]}
    except Exception:
        return None


def cpu_baseline_retrieval(args, feats_h):
    """The reference's NumPy op sequence (oracle port) on a bounded query sample, host cores."""
    from oracle import retrieval_oracle as ro
    qn = min(args.cpu_sample_queries, feats_h.shape[0])
    f = feats_h.copy()
    t0 = time.perf_counter()
    rank = ro.pairwise_retrieval_numpy(f, normalize=(args.metric == "cosine"), queries=slice(0, qn))
    dt = time.perf_counter() - t0
    assert rank.shape == (qn, feats_h.shape[0])
    return {"value": qn * feats_h.shape[0] / dt / 1e6, "unit": "Mpairs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d of %d queries x %d gallery, D=%d: np.linalg.norm + np.dot (BLAS threads = all cores) + "
                      "np.argsort(kind='stable') (single-threaded in NumPy), %.1f s" % (qn, feats_h.shape[0], feats_h.shape[0], feats_h.shape[1], dt)}


def main():
    args = parse()
    rank, world, _ = init_dist(args)
    if args.workload == "train":
        from train_bench import bench_train
        out = bench_train(args, rank, world)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from train_bench import cpu_baseline_train
            out["cpu_baseline"] = cpu_baseline_train(args)
    else:
        out, feats_h = bench_retrieval(args, rank, world)
        if args.with_train:
            try:
                from train_bench import bench_train
                out["train"] = bench_train(args, rank, world)
            except Exception as e:   # the retrieval line must survive a training-side failure
                out["train"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported baselines: rank 0 at N = 1 only, bounded samples
            out["cpu_baseline"] = cpu_baseline_retrieval(args, feats_h)
            if args.with_train and "error" not in out["train"]:
                try:
                    from train_bench import cpu_baseline_train
                    out["train"]["cpu_baseline"] = cpu_baseline_train(args)
                except Exception as e:
                    out["train"]["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
