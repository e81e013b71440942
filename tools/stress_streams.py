#!/usr/bin/env python
"""Every retrieval / loss entry point under two-stream contention: result of a call made while a second stream keeps the CUs busy
== result of the same call made alone, bit for bit.

Round 4 found se_topk_rows returning unsorted rows whenever another stream's kernels shared its CUs (a work-group barrier compiled
without the LDS wait); no single-stream test could see that.  This runs each entry point `iters` times with
three kinds of side work (distance tiles: MFMA + LDS; rankings: LDS atomics; top-k selects: barrier-dense) and compares with the
solo result.  Solo results themselves are what the GPU parity tests hold against the oracle.

    python tools/stress_streams.py [iters]          # exit 1 on any difference
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "semantic-embeddings_amd"), ROOT):
    sys.path.insert(0, p)
import sehip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
f32 = lambda *s: rng.standard_normal(s).astype(np.float32)

# ---- inputs ----
g20 = dev(f32(20000, 100)); sehip.normalize_rows_(g20)
g3 = dev(f32(3001, 200)); sehip.normalize_rows_(g3)
gw = dev(f32(20000, 320)); sehip.normalize_rows_(gw)              # padded width 384: the 256 x 256 LDS-DMA filter kernel
raw = dev(f32(8192, 1000))
sq20 = sehip.row_sqnorm(g20)
pd_short = sehip.pairwise_dist(g3[:512], g3)                       # 3001 columns: short-row instantiations
pd_mid = sehip.pairwise_dist(g20[:1024], g20)                      # 20,000 columns
gl = dev(f32(60000, 64)); sehip.normalize_rows_(gl)
pd_long = sehip.pairwise_dist(gl[:256], gl)                        # 60,000 columns: two sorted runs + merge
C = 100
cls = dev(rng.integers(0, C, size=20000).astype(np.int32))
tab = rng.random((C, C)); tab = (tab + tab.T) / 2; np.fill_diagonal(tab, 1.0)
counts = np.bincount(cls.cpu().numpy(), minlength=C)
best = np.stack([np.cumsum(np.repeat(tab[c][np.argsort(-tab[c], kind="stable")], counts[np.argsort(-tab[c], kind="stable")])) for c in range(C)])
tab_d, best_d = dev(tab), dev(best)
rk_mid = sehip.rank_rows(pd_mid)
ks = torch.arange(1, 251, dtype=torch.int32, device="cuda")
qidx = torch.arange(1024, dtype=torch.int32, device="cuda")
curves = sehip.hprec_reciprocal_curves(best_d, best_d)
emb = dev(f32(C, 100)); sehip.normalize_rows_(emb)
xb = dev(f32(4096, 100)); yb = dev(rng.integers(0, C, size=4096).astype(np.int64))
parts = torch.stack([torch.stack([sehip.topk_rows(pd_mid[:, i * 2500:(i + 1) * 2500].contiguous(), 251, col_offset=i * 2500)[j].view(torch.int32)
                                  for j in (0, 1)]) for i in range(8)])     # [8, 2, Q, k] packed lists


def as_tuple(r):
    return tuple(r) if isinstance(r, (tuple, list)) else (r,)


OPS = {
    "normalize_rows_": lambda: sehip.normalize_rows_(raw.clone()),
    "row_sqnorm": lambda: sehip.row_sqnorm(raw),
    "pairwise_dist cosine symmetric": lambda: sehip.pairwise_dist(g20[:4096], g20[:4096]),
    "pairwise_dist Euclid general": lambda: sehip.pairwise_dist(g20[:2048], g20, metric=sehip.METRIC_EUCLID, sqa=sq20[:2048], sqb=sq20),
    "rank_rows 3,001 columns": lambda: sehip.rank_rows(pd_short),
    "rank_rows 20,000 columns": lambda: sehip.rank_rows(pd_mid),
    "rank_rows 60,000 columns (runs + merge)": lambda: sehip.rank_rows(pd_long),
    "topk_rows k=251": lambda: sehip.topk_rows(pd_mid, 251),
    "topk_rows k=1000": lambda: sehip.topk_rows(pd_mid, 1000),
    "topk_merge packed 8 parts": lambda: sehip.topk_merge(parts),
    "retrieve_topk fused cosine": lambda: sehip.retrieve_topk(g20[:4096], g20, 251),
    "retrieve_topk fused Euclid": lambda: sehip.retrieve_topk(g20[:4096], g20, 100, metric=sehip.METRIC_EUCLID, sqq=sq20[:4096], sqg=sq20),
    "retrieve_topk fused, long rows (LDS-DMA)": lambda: sehip.retrieve_topk(gw[:4096], gw, 251),
    "retrieve_topk slab (3,001 rows)": lambda: sehip.retrieve_topk(g3[:512], g3, 64),
    "hierarchical_precision": lambda: sehip.hierarchical_precision(rk_mid, cls, cls[:1024].contiguous(), qidx, tab_d, tab_d, best_d, best_d, ks,
                                                                     ahp_len=0, want_ap=True, curves=curves),
    "cosine_loss fwd+bwd": lambda: (sehip.cosine_loss_forward(xb, yb, emb)[0], sehip.cosine_loss_backward(xb, yb, emb)),
    "nn_accuracy": lambda: sehip.nn_accuracy(xb, yb, emb, dot_prod_sim=True, k=5),
    "devise_ranking_loss": lambda: sehip.devise_ranking_loss(xb, yb, emb),
}
SIDE = {
    "distance tiles": lambda: sehip.pairwise_dist(g20[:6000], g20[:6000]),
    "rankings": lambda: sehip.rank_rows(pd_mid[:256]),
    "top-k selects": lambda: sehip.topk_rows(pd_mid[:512], 251),
}

side = torch.cuda.Stream()
failed = 0
for name, op in OPS.items():
    torch.cuda.synchronize()
    want = [t.clone() for t in as_tuple(op()) if torch.is_tensor(t)]
    torch.cuda.synchronize()
    again = [t for t in as_tuple(op()) if torch.is_tensor(t)]
    torch.cuda.synchronize()
    if not all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(want, again)):
        print("%-42s NOT deterministic alone -- skipped" % name); continue
    line = "%-42s" % name
    for sname, noise in SIDE.items():
        bad = 0
        for it in range(iters):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    noise()
            got = [t for t in as_tuple(op()) if torch.is_tensor(t)]
            bad += not all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(want, got))
        torch.cuda.synchronize()
        line += "  %s: %d/%d differ" % (sname, bad, iters)
        failed += bad
    print(line, flush=True)
print("stress_streams: %d differing calls" % failed)
sys.exit(1 if failed else 0)
