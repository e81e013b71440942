"""oracle/verify.py -- TEST INFRASTRUCTURE ONLY: full-size checks of one retrieval step against the oracle.

Used by tests/test_gpu_retrieval.py (the benchmarked 50,000 x 50,000 x 100 configuration) and by bench.py AFTER its
timed region (``"verified"`` in the bench line) -- as the checker, never as the thing measured.

What is checked on device results ``pd`` [Q, N] f32 (distances) and ``rk`` [Q, N] int32 / int64 (ranking) that the HIP path
produced from the feature rows ``feats`` (exactly as the distance kernel saw them: normalised for the cosine branch):

* size-independent properties of EVERY row, with torch ops on the device: the ranking is a permutation of 0..N-1, the
  distances are non-decreasing along it, indices ascend inside runs of equal distance (the canonical tie rule), and -- when
  the step is the all-pairs case Q == N -- the distance matrix equals its transpose bit for bit (evaluate_retrieval.py:59's
  ``ssyrk`` result is exactly symmetric; the symmetric kernel mirrors upper-triangle tiles);
* sampled rows (first / last tile rows, rows on both sides of every kind of tile boundary, the last row, random rows):
  distances == oracle/canon.c's FMA chain bit for bit and ranking == the canonical (distance, index) sort of that row.
"""
import numpy as np

from oracle import retrieval_oracle as ro


def sample_rows(q, n_random=40, tile=128, seed=123):
    """Rows on tile boundaries of the distance kernel's 128 x 128 tiling + random rows (sorted, unique)."""
    last_tile = (q - 1) // tile * tile
    fixed = [0, 1, tile - 1, tile, tile + 1, 2 * tile - 1, 2 * tile, q // 2, q // 2 + 1, last_tile - 1, last_tile, last_tile + 1,
             q - 2, q - 1]
    mid_tile = (q // 2) // tile * tile
    fixed += [mid_tile - 1, mid_tile, mid_tile + tile - 1, mid_tile + tile]
    rng = np.random.default_rng(seed)
    rows = set(int(r) for r in fixed if 0 <= r < q) | set(int(r) for r in rng.integers(0, q, size=n_random))
    return sorted(rows)


def check_properties(pd, rk, symmetric, chunk=4096):
    """Every-row properties on the device; returns a dict of booleans (all must be True)."""
    import torch
    q, n = pd.shape
    ok = {"permutation": True, "sorted": True, "tie_order": True}
    ar = torch.ones((1,), dtype=torch.uint8, device=pd.device)
    for r0 in range(0, q, chunk):
        r1 = min(q, r0 + chunk)
        idx = rk[r0:r1].long()
        ok["permutation"] &= bool(((idx >= 0) & (idx < n)).all())
        if not ok["permutation"]:
            break
        seen = torch.zeros((r1 - r0, n), dtype=torch.uint8, device=pd.device)
        seen.scatter_(1, idx, ar.expand(r1 - r0, n))
        ok["permutation"] &= bool(seen.all())
        srt = torch.gather(pd[r0:r1], 1, idx)
        ok["sorted"] &= bool((srt[:, 1:] >= srt[:, :-1]).all())
        ok["tie_order"] &= bool(((srt[:, 1:] > srt[:, :-1]) | (idx[:, 1:] > idx[:, :-1])).all())
        del seen, srt, idx
    if symmetric:
        sym = True
        for r0 in range(0, q, chunk):
            r1 = min(q, r0 + chunk)
            sym &= bool(torch.equal(pd[r0:r1, :], pd[:, r0:r1].t()))
        ok["symmetric"] = sym
    return ok


def check_sampled_rows(feats, pd, rk, metric, rows, queries=None, kblocks=None):
    """Bit-exact comparison of the sampled rows with the canonical oracle.  ``feats``: host f32 [N, D] gallery rows as the
    distance kernel saw them; ``queries``: host rows of the left operand (default: the gallery itself)."""
    import torch
    rows = list(rows)
    qf = feats if queries is None else queries
    want_pd = ro.canon_pdist(np.ascontiguousarray(qf[rows]), feats, metric, kblocks)
    ridx = torch.as_tensor(rows, device=pd.device)
    got_pd = pd[ridx].cpu().numpy()
    got_rk = rk[ridx].cpu().numpy().astype(np.int64)
    want_rk = ro.canon_rank_rows(want_pd).astype(np.int64)
    return {"sampled_distances": bool(np.array_equal(got_pd, want_pd)), "sampled_rankings": bool(np.array_equal(got_rk, want_rk)),
            "rows_checked": len(rows)}


def verify_retrieval_step(feats, pd, rk, metric, queries=None, n_random=40, kblocks=None):
    """All checks; returns (all_ok, detail dict)."""
    symmetric = queries is None and pd.shape[0] == pd.shape[1]
    detail = check_properties(pd, rk, symmetric)
    detail.update(check_sampled_rows(feats, pd, rk, metric, sample_rows(pd.shape[0], n_random), queries, kblocks))
    all_ok = all(v for k, v in detail.items() if k != "rows_checked")
    return all_ok, detail


def verify_topk_sample(queries, gallery, metric, k, got_d, got_i, rows, col_offset=0, kblocks=None):
    """Sampled queries of a fused distance + top-k call (``se_retrieve_topk``: one shard of the sharded-gallery split, or an
    all-pairs call) against the oracle: rows ``rows`` of ``(got_d, got_i)`` [Q, k] must equal the first k entries of the canonical
    ranking of canon.c's distances (same K-block list) bit for bit.  ``queries`` / ``gallery``: host f32 rows as the kernel saw them."""
    rows = [int(r) for r in rows]
    want_pd = ro.canon_pdist(np.ascontiguousarray(queries[rows]), gallery, metric, kblocks)
    wd, wi = ro.canon_topk_rows(want_pd, k, col_offset=col_offset)
    gd = np.asarray(got_d)[rows] if not hasattr(got_d, "cpu") else got_d[rows].cpu().numpy()
    gi = np.asarray(got_i)[rows] if not hasattr(got_i, "cpu") else got_i[rows].cpu().numpy()
    bad_rows = [rows[j] for j in range(len(rows)) if not (np.array_equal(gd[j], wd[j]) and np.array_equal(gi[j], wi[j]))]
    return {"queries_checked": len(rows), "distances_bit_equal": bool(np.array_equal(gd, wd)), "indices_equal": bool(np.array_equal(gi, wi)),
            "mismatching_queries": bad_rows[:8]}
