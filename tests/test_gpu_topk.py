"""GPU parity tests of the fused distance + top-k (se_retrieve_topk): HIP kernels through the C ABI vs the oracle and vs
the heads of the imported reference's rankings (tests/golden/topk_head_*.npz, D = 555 / 1000 with the BLAS K-block list).

Bar: out_i / out_d == the first k entries of the canonical full ranking, bit for bit (reference:
evaluate_retrieval.py:57-67), on every path of the driver: distance slab (small galleries), fused sample / filter / sort
passes, and the exact per-query fallback.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as ro

pytestmark = pytest.mark.gpu

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT_DIR, 'semantic-embeddings_amd')
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "topk_head_*.npz")))


@pytest.fixture(scope="module")
def sehip():
    import sehip as m
    m.lib()
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def want_topk(q, g, k, metric, kblocks=None, col_offset=0):
    pd = ro.canon_pdist(q, g, metric, kblocks=kblocks)
    return ro.canon_topk_rows(pd, k, col_offset=col_offset)


def check_against_reference_head(sehip_mod, path, k=251):
    feat, norm, kb, head = ro.load_topk_fixture(path)
    x = dev(feat.copy())
    if norm:
        sehip_mod.normalize_rows_(x)
    metric = ro.METRIC_COSINE if norm else ro.METRIC_EUCLID
    d, i = sehip_mod.retrieve_topk(x, x, k, metric=metric, kblocks=kb)
    xh = x.cpu().numpy()
    pd = ro.canon_pdist(xh, None, metric, kblocks=kb)
    wd, wi = ro.canon_topk_rows(pd, k)
    # gate 1: the canonical oracle, bit for bit
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(d.cpu().numpy(), wd)
    # gate 2: the head of the imported reference's ranking, except inside exact-tie groups
    got = i.cpu().numpy().astype(np.int64)
    for r in np.nonzero((got != head[:, :k]).any(axis=1))[0]:
        assert np.array_equal(pd[r][got[r]], pd[r][head[r, :k]]), "row %d differs outside a tie group" % r
    # and the single-chain arithmetic would NOT have reproduced the reference on this fixture
    d1, i1 = sehip_mod.retrieve_topk(x, x, k, metric=metric)
    assert not np.array_equal(i1.cpu().numpy(), wi) or not np.array_equal(d1.cpu().numpy(), wd)


@pytest.mark.parametrize("path", GOLDEN)
def test_topk_reference_heads_with_kblocks(sehip, path):
    """640-item galleries take the distance-slab path of the product library."""
    check_against_reference_head(sehip, path)


def run_with_tuning_lib(body, env=None, timeout=900):
    """Run `body` in a subprocess bound to libsehip_tuning.so (path / threshold switches exist only there)."""
    code = (
        "import os, sys, glob, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_topk as T\n"
    ) % ([PKG_DIR, ROOT_DIR], os.path.dirname(__file__)) + body + "\nprint('subprocess-ok')\n"
    e = dict(os.environ, SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert out.returncode == 0 and "subprocess-ok" in out.stdout, out.stdout[-4000:]
    return out.stdout


def test_topk_reference_heads_through_the_fused_passes():
    """The same fixtures with the fused path pinned (SE_TOPK_FUSED=1: sample pass over half the gallery, threshold, filter
    pass, list sort) -- K-block arithmetic inside the MFMA tile loop of the fused kernels."""
    run_with_tuning_lib(
        "for p in T.GOLDEN:\n"
        "    T.check_against_reference_head(sehip, p)\n", env={"SE_TOPK_FUSED": "1"})


FUSED_CASES = (
    # q, n, d, k, metric, kblocks, col_offset
    "CASES = [(300, 3000, 100, 25, 0, None, 0), (77, 20000, 100, 251, 0, None, 1000), (130, 2999, 64, 40, 1, None, 7),\n"
    "         (200, 4097, 7, 10, 0, None, 0), (65, 2500, 555, 100, 1, [278, 277], 0), (129, 2048, 1000, 251, 0, [448, 276, 276], 5),\n"
    "         (64, 1500, 130, 1, 1, None, 0), (40, 9000, 200, 1024, 0, None, 0)]\n"
    "def run_cases(label):\n"
    "    for (q, n, d, k, metric, kb, off) in CASES:\n"
    "        rng = np.random.default_rng(q + n + d)\n"
    "        g = rng.standard_normal((n, d)).astype(np.float32)\n"
    "        qs = np.ascontiguousarray(g[rng.permutation(n)[:q]]) if q <= n else rng.standard_normal((q, d)).astype(np.float32)\n"
    "        if metric == 0:\n"
    "            g = ro.canon_normalize_rows(g); qs = ro.canon_normalize_rows(qs)\n"
    "        dd, ii = sehip.retrieve_topk(torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda(), k, metric=metric, kblocks=kb, col_offset=off)\n"
    "        wd, wi = T.want_topk(qs, g, k, metric, kb, off)\n"
    "        assert np.array_equal(ii.cpu().numpy(), wi), (label, q, n, d, k)\n"
    "        assert np.array_equal(dd.cpu().numpy(), wd), (label, q, n, d, k)\n"
)


def test_fused_topk_equals_head_of_canonical_ranking():
    """Fused passes on ragged shapes (partial tiles in both directions, unaligned D, K-blocks, both metrics, k = 1 ... 1024)."""
    run_with_tuning_lib(FUSED_CASES + "os.environ['SE_TOPK_FUSED'] = '1'\nrun_cases('fused')\n")


def test_fused_topk_big_tile_filter_kernel_on_ragged_shapes():
    """The 256 x 256 filter kernel (product default for padded widths >= 256 on large problems) pinned on SMALL ragged ones
    (SE_PF_BIG=1): partial tiles in both directions, one-tile galleries, several K-chunks, K-blocks, both metrics, and the same
    cases with capacities forced small (its overflow marking).  Its product-size run is test_product_library_d1000_kblocks_large_gallery."""
    cases = ("CASES = [(300, 3000, 100, 25, 0, None, 0), (77, 20000, 300, 251, 0, None, 1000), (130, 2999, 64, 40, 1, None, 7),\n"
             "         (513, 4097, 260, 10, 0, None, 0), (65, 2500, 555, 100, 1, [278, 277], 0), (129, 2048, 1000, 251, 0, [448, 276, 276], 5),\n"
             "         (257, 700, 130, 1, 1, None, 0), (40, 9000, 200, 500, 0, None, 0)]\n")
    body = FUSED_CASES[FUSED_CASES.index("def run_cases"):]
    run_with_tuning_lib(cases + body + "os.environ['SE_TOPK_FUSED'] = '1'\nos.environ['SE_PF_BIG'] = '1'\nrun_cases('big tiles')\n"
                        "os.environ['SE_TOPK_CAP'] = '256'\nCASES = [c for c in CASES if c[3] <= 128]\nrun_cases('big tiles, overflow')\n")
    # irregular rows (NaN / inf / huge / zero / denormal) in gallery and queries through the same kernel: NaN accumulators pass its compare
    run_with_tuning_lib(
        "os.environ['SE_TOPK_FUSED'] = '1'\nos.environ['SE_PF_BIG'] = '1'\n"
        "rng = np.random.default_rng(34)\n"
        "n, d, k = 5000, 260, 60\n"
        "g = rng.standard_normal((n, d)).astype(np.float32)\n"
        "g[17, 3] = np.inf; g[18, 5] = -np.inf; g[19, 0] = np.nan; g[20] = 0.0; g[21, 7] = 3e30; g[22, 1] = -2e19; g[23] *= 1e-41\n"
        "g[24, 3] = np.inf; g[24, 4] = -np.inf; g[4100, 259] = np.nan\n"
        "qs = np.concatenate([g[[0, 1, 17, 18, 19, 20, 21, 22, 23, 24, 4999]], g[300:600]]).copy()\n"
        "qs[0, 2] = 1e-42\n"
        "for metric in (0, 1):\n"
        "    with np.errstate(all='ignore'):\n"
        "        wd, wi = T.want_topk(qs, g, k, metric)\n"
        "    dd, ii = sehip.retrieve_topk(torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda(), k, metric=metric)\n"
        "    assert np.array_equal(ii.cpu().numpy(), wi), metric\n"
        "    assert np.array_equal(dd.cpu().numpy(), wd, equal_nan=True), metric\n")


def test_fused_topk_exact_fallback_paths():
    """Thresholds forced too low (j = 1: lists shorter than k) and capacities forced too small (overflow): every query is
    flagged and redone by the exact kernel (VALU FMA chain + radix select) -- same bits."""
    run_with_tuning_lib(FUSED_CASES +
                        "os.environ['SE_TOPK_FUSED'] = '1'\n"
                        "os.environ['SE_TOPK_J'] = '1'\nrun_cases('short lists')\n"
                        "del os.environ['SE_TOPK_J']\nos.environ['SE_TOPK_CAP'] = '256'\nCASES = [c for c in CASES if c[3] <= 128]\nrun_cases('overflow')\n")


def test_fused_topk_degenerate_rows():
    """Rows the sample cannot resolve: duplicated gallery items (tie groups straddling rank k), all-equal distances, NaN
    queries (every distance NaN: sorted last, index order), a zero query under the Euclidean metric."""
    run_with_tuning_lib(
        "os.environ['SE_TOPK_FUSED'] = '1'\n"
        "rng = np.random.default_rng(4)\n"
        "n, d, k = 6000, 48, 100\n"
        "g = rng.standard_normal((n, d)).astype(np.float32)\n"
        "g[1000:1800] = g[10]\n"                       # 800 duplicates of one item
        "g[3000:3050] = 0.0\n"
        "qs = g[[10, 11, 1000, 3000, 5999]].copy()\n"
        "qs = np.concatenate([qs, np.full((1, d), np.nan, np.float32), np.zeros((1, d), np.float32)])\n"
        "for metric in (0, 1):\n"
        "    dd, ii = sehip.retrieve_topk(torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda(), k, metric=metric)\n"
        "    wd, wi = T.want_topk(qs, g, k, metric)\n"
        "    assert np.array_equal(ii.cpu().numpy(), wi), metric\n"
        "    assert np.array_equal(dd.cpu().numpy(), wd, equal_nan=True), metric\n"
        "g[:] = 1.0\n"                                  # one distance value everywhere
        "dd, ii = sehip.retrieve_topk(torch.from_numpy(qs[:3]).cuda(), torch.from_numpy(g).cuda(), k, metric=1)\n"
        "wd, wi = T.want_topk(qs[:3], g, k, 1)\n"
        "assert np.array_equal(ii.cpu().numpy(), wi) and np.array_equal(dd.cpu().numpy(), wd, equal_nan=True)\n")


@pytest.mark.parametrize("metric", [0, 1])
def test_product_path_takes_fused_kernels_from_16384_rows(sehip, metric):
    """The product library's own path choice (no switches): a 20,000-row gallery runs the fused passes."""
    rng = np.random.default_rng(8 + metric)
    n, d, q, k = 20000, 100, 384, 251
    g = rng.standard_normal((n, d)).astype(np.float32)
    if metric == 0:
        g = ro.canon_normalize_rows(g)
    qs = np.ascontiguousarray(g[:q])
    sehip.phase_timing(True)                      # the library's own phase events name the path that ran
    try:
        dd, ii = sehip.retrieve_topk(dev(qs), dev(g), k, metric=metric, col_offset=123)
        phases, counters = sehip.phase_timing_read()
    finally:
        sehip.phase_timing(False)
    assert {"convert", "sample", "filter", "refine"} <= set(phases), "fused pre-filter path expected (candidate lists, not a distance slab): %s" % phases
    assert counters is not None and counters["queries"] == q and counters["redone"] == 0 and counters["candidates"] >= q * k
    wd, wi = want_topk(qs, g, k, metric, None, 123)
    assert np.array_equal(ii.cpu().numpy(), wi)
    assert np.array_equal(dd.cpu().numpy(), wd)


@pytest.mark.parametrize("metric", [0, 1])
def test_class_sorted_gallery_stays_on_the_fast_path(sehip, metric):
    """A gallery sorted by class (ILSVRC training features come that way) puts every query's neighbours into a few adjacent gallery
    tiles, i.e. into two or three of its candidate sub-lists.  Round 4 sized the sub-lists for shuffled galleries: they overflowed and
    every such query went to the exact fallback (52 ms instead of 3 ms at 50k x 50k, 1.45 s instead of 12 ms on an ILSVRC-sized shard).
    The sub-lists now share a spill region: the heads are the oracle's, and NO query is redone."""
    rng = np.random.default_rng(90 + metric)
    n, d, C, q, k = 24000, 96, 48, 640, 251
    cen = rng.standard_normal((C, d)).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    y = np.sort(rng.integers(0, C, size=n))
    g = (cen[y] + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    if metric == 0:
        g = ro.canon_normalize_rows(g)
    rows = np.linspace(0, n - 1, q).astype(np.int64)
    qs = np.ascontiguousarray(g[rows])
    sehip.phase_timing(True)
    try:
        dd, ii = sehip.retrieve_topk(dev(qs), dev(g), k, metric=metric)
        _, counters = sehip.phase_timing_read()
    finally:
        sehip.phase_timing(False)
    wd, wi = want_topk(qs, g, k, metric, None, 0)
    assert np.array_equal(ii.cpu().numpy(), wi)
    assert np.array_equal(dd.cpu().numpy(), wd)
    assert counters["redone"] == 0, counters
    # the all-pairs form of the same gallery
    x = dev(g)
    sq = sehip.row_sqnorm(x) if metric == 1 else None
    sehip.phase_timing(True)
    try:
        d2, i2 = sehip.retrieve_topk(x, x, k, metric=metric, sqq=sq, sqg=sq)
        _, counters = sehip.phase_timing_read()
    finally:
        sehip.phase_timing(False)
    assert np.array_equal(i2[rows].cpu().numpy(), wi) and np.array_equal(d2[rows].cpu().numpy(), wd)
    assert counters["redone"] == 0, counters


def test_fused_topk_full_size_head_equals_full_ranking(sehip):
    """BASELINE configs[2] size: top-251 of 50,000 x 50,000 x 100 through the fused kernels == the first 251 columns of
    se_rank_rows(se_pairwise_dist) on a 2,048-query slice, and == the oracle on sampled rows."""
    n, d, k = 50000, 100, 251
    x = dev(np.random.default_rng(0).standard_normal((n, d)).astype(np.float32))
    sehip.normalize_rows_(x)
    dd, ii = sehip.retrieve_topk(x, x, k, metric=ro.METRIC_COSINE)
    pd = sehip.pairwise_dist(x[:2048], x, metric=ro.METRIC_COSINE)
    rk = sehip.rank_rows(pd)
    assert bool((rk[:, :k] == ii[:2048]).all())
    assert bool((torch.gather(pd, 1, rk[:, :k].long()) == dd[:2048]).all())
    assert bool((ii[:, 0] == torch.arange(n, device="cuda", dtype=torch.int32)).all())     # every query finds itself first
    xh = x.cpu().numpy()
    rows = [0, 127, 128, 25000, 49999]
    wd, wi = want_topk(xh[rows], xh, k, ro.METRIC_COSINE)
    assert np.array_equal(ii[rows].cpu().numpy(), wi) and np.array_equal(dd[rows].cpu().numpy(), wd)


@pytest.mark.parametrize("metric", [0, 1])
def test_product_library_d1000_kblocks_large_gallery(sehip, metric):
    """BASELINE configs[4]'s arithmetic through the PRODUCT library (no switches): D = 1000 with the host-BLAS K-block list
    [448, 276, 276] (what np.dot computes at this depth, evaluate_retrieval.py:59), a gallery above the 16,384-row threshold of
    the fused passes, k = 251, both metrics, trained-like features (class-embedding row + noise: dense near-ties) -- sampled
    queries bit-equal to canon.c with the same list."""
    from evaluate_retrieval import host_blas_kblocks
    from oracle import verify
    kb = host_blas_kblocks(1000)
    assert kb == [448, 276, 276]
    n, q, d, k = 18000, 640, 1000, 251
    emb = np.load(os.path.join(os.path.dirname(__file__), "golden", "imagenet_mintree_unitsphere.npz"))["embedding"]
    rng = np.random.default_rng(31 + metric)
    y = rng.integers(0, emb.shape[0], size=n)
    g = (emb[y] + 0.03 * rng.standard_normal((n, d))).astype(np.float32)
    yq = rng.integers(0, emb.shape[0], size=q)
    qs = (emb[yq] + 0.03 * rng.standard_normal((q, d))).astype(np.float32)
    if metric == 0:
        g, qs = ro.canon_normalize_rows(g), ro.canon_normalize_rows(qs)
    sehip.phase_timing(True)                      # the library's own phase events name the path that ran
    try:
        dd, ii = sehip.retrieve_topk(dev(qs), dev(g), k, metric=metric, kblocks=kb, col_offset=n)
        phases, _ = sehip.phase_timing_read()
    finally:
        sehip.phase_timing(False)
    assert {"convert", "sample", "filter", "refine"} <= set(phases), "fused pre-filter path expected (candidate lists, not a distance slab): %s" % phases
    rows = verify.sample_rows(q, n_random=12)
    det = verify.verify_topk_sample(qs, g, metric, k, dd, ii, rows, col_offset=n, kblocks=kb)
    assert det["indices_equal"] and det["distances_bit_equal"], det
    # every list sorted under the canonical order, indices inside the shard's global range
    assert bool(((dd[:, 1:] > dd[:, :-1]) | ((dd[:, 1:] == dd[:, :-1]) & (ii[:, 1:] > ii[:, :-1]))).all())
    assert bool(((ii >= n) & (ii < 2 * n)).all())
    # all-pairs call on the same gallery (upper-triangle walk + mirrored filter) with the list
    x = dev(g)
    sq = sehip.row_sqnorm(x) if metric == 1 else None
    d2, i2 = sehip.retrieve_topk(x, x, k, metric=metric, kblocks=kb, sqq=sq, sqg=sq)
    rows2 = verify.sample_rows(n, n_random=6)
    det2 = verify.verify_topk_sample(g, g, metric, k, d2, i2, rows2, kblocks=kb)
    assert det2["indices_equal"] and det2["distances_bit_equal"], det2


def test_topk_merge_packed_equals_separate_lists(sehip):
    """se_topk_merge_packed on the receive buffer of ONE all-gather ([parts, 2, q, k]: distance bits | indices per part) ==
    se_topk_merge on separate [parts, q, k] tensors == the oracle's merge."""
    rng = np.random.default_rng(5)
    parts, q, k = 4, 300, 251
    d = np.sort(rng.standard_normal((parts, q, k)).astype(np.float32), axis=-1)
    d[1, :, :40] = d[0, :, :40]                              # exact ties across parts: global index decides
    i = (np.arange(parts)[:, None, None] * 100000 + np.sort(rng.integers(0, 100000, size=(parts, q, k)), axis=-1)).astype(np.int32)
    packed = np.stack([d.view(np.int32), i], axis=1)         # [parts, 2, q, k]
    md, mi = sehip.topk_merge(dev(d), dev(i))
    pd_, pi_ = sehip.topk_merge(dev(packed))
    wd, wi = ro.canon_topk_merge(d, i)
    assert np.array_equal(md.cpu().numpy(), wd) and np.array_equal(mi.cpu().numpy(), wi)
    assert np.array_equal(pd_.cpu().numpy(), wd) and np.array_equal(pi_.cpu().numpy(), wi)


@pytest.mark.parametrize("parts,k", [(1, 251), (2, 1), (8, 251), (3, 64), (5, 65), (4, 128), (7, 300), (8, 512), (3, 1024), (2, 1500)])
def test_topk_merge_wave_kernel_shapes(sehip, parts, k):
    """The wave-per-query merge (k <= 1024: running best list in registers, one bitonic merge per part) and the LDS sort it replaced
    (k > 1024) against the oracle: every register count (k = 1 ... 1024), ties across parts, +inf padding of short shards, NaN last."""
    rng = np.random.default_rng(parts * 10007 + k)
    q = 257
    d = np.sort(rng.standard_normal((parts, q, k)).astype(np.float32), axis=-1)
    i = (np.arange(parts)[:, None, None] * 1000000 + np.sort(rng.integers(0, 1000000, size=(parts, q, k)), axis=-1)).astype(np.int32)
    if parts > 1:
        d[1, :, :k // 3] = d[0, :, :k // 3]                       # exact ties across parts: the global index decides
        d[-1, ::3, k - k // 4:] = np.inf                          # a short shard's padding
        i[-1, ::3, k - k // 4:] = 2 ** 31 - 1
    d[0, 5, k - 1] = np.nan                                       # NaN sorts last
    md, mi = sehip.topk_merge(dev(d), dev(i))
    wd, wi = ro.canon_topk_merge(d, i)
    assert np.array_equal(mi.cpu().numpy(), wi)
    assert np.array_equal(md.cpu().numpy(), wd, equal_nan=True)


def test_topk_merge_accepts_unsorted_parts(sehip):
    """The C ABI does not promise sorted parts to se_topk_merge: a part that is not ascending is sorted inside the wave first."""
    rng = np.random.default_rng(77)
    parts, q, k = 4, 130, 251
    d = rng.standard_normal((parts, q, k)).astype(np.float32)                       # unsorted
    d[2] = np.sort(d[2], axis=-1)                                                   # one part sorted, the others not
    i = rng.permutation(parts * q * k).reshape(parts, q, k).astype(np.int32)
    md, mi = sehip.topk_merge(dev(d), dev(i))
    wd, wi = ro.canon_topk_merge(d, i)
    assert np.array_equal(mi.cpu().numpy(), wi) and np.array_equal(md.cpu().numpy(), wd)


# ---------------------------------------------------------------- bf16 pre-filter (prefilter.hip + pf_refine_kernel)

def test_fp32_fused_passes_stay_covered():
    """SE_TOPK_PREFILTER=0 (tuning build) pins the fp32 form of the fused passes (what k > 512 takes in the product): reference heads
    with K-blocks and the ragged cases, as before the bf16 pre-filter existed."""
    run_with_tuning_lib(
        "for p in T.GOLDEN:\n"
        "    T.check_against_reference_head(sehip, p)\n", env={"SE_TOPK_FUSED": "1", "SE_TOPK_PREFILTER": "0"})
    run_with_tuning_lib(FUSED_CASES + "os.environ['SE_TOPK_FUSED'] = '1'\nos.environ['SE_TOPK_PREFILTER'] = '0'\nrun_cases('fp32 fused')\n")


PROBE = (
    "import ctypes\n"
    "from sehip._lib import lib, ptr, check\n"
    "def probe(qs, g, metric, nkb=1):\n"
    "    L = lib(); f = L.se_tuning_prefilter_probe\n"
    "    i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int\n"
    "    f.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, ci, ci, vp, i64, vp, vp, i64, vp]\n"
    "    q, d = qs.shape; n = g.shape[0]; kp = (d + 127) // 128 * 128\n"
    "    Q, G = torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda()\n"
    "    sq = sehip.row_sqnorm(Q) if metric == 1 else None; sg = sehip.row_sqnorm(G) if metric == 1 else None\n"
    "    out = torch.empty((n, q), dtype=torch.float32, device='cuda'); eps = torch.empty((q,), dtype=torch.float32, device='cuda')\n"
    "    ws = torch.empty((2 * (n + q) * kp + 16 * (n + q) + 8 * q + 8192,), dtype=torch.uint8, device='cuda')\n"
    "    check(f(ptr(Q), d, ptr(G), d, ptr(sq), ptr(sg), q, n, d, metric, nkb, ptr(out), q, ptr(eps), ptr(ws), ws.numel(),\n"
    "            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'probe')\n"
    "    torch.cuda.synchronize()\n"
    "    return out.cpu().numpy(), eps.cpu().numpy()\n"
    "def image(x):\n"
    "    m = float(np.abs(x).max()); e = 14 - (np.frexp(m)[1] if m > 0 else 14)\n"
    "    xs = (x.astype(np.float64) * 2.0 ** e).astype(np.float32)\n"
    "    h = xs.astype(np.float16); h[np.abs(xs) < 2.0 ** -14] = 0\n"
    "    return h.astype(np.float64) * 2.0 ** -e\n"
)


def test_prefilter_error_bound_holds_on_hardware():
    """The pre-filter's correctness rests on |d~ - d| <= eps(query) for EVERY pair.  eps is derived from the operands' actual bf16
    rounding residuals plus assumption A1 about the matrix core's accumulation (|error| <= 2^-20 (|C| + sum |products|) per
    v_mfma_f32_32x32x16_f16): measured here on gaussian, clustered, wide-dynamic-range and cancellation-heavy operands, both metrics,
    D = 100 / 555 / 1000 -- (a) the bound holds with room, (b) A1 itself: the matrix core's result against the float64 dot product of
    the SAME fp16 images is within 1/8 of what A1 allows."""
    out = run_with_tuning_lib(
        PROBE +
        "rng = np.random.default_rng(12)\n"
        "worst_bound = worst_a1 = 0.0\n"
        "for d in (100, 555, 1000):\n"
        "    n, q = 700, 200\n"
        "    base = rng.standard_normal((n, d)).astype(np.float32)\n"
        "    cases = {'gauss': base,\n"
        "             'clustered': (base[rng.integers(0, 12, size=n)] + 0.02 * rng.standard_normal((n, d))).astype(np.float32),\n"
        "             'dynamic': (base * np.exp2(rng.integers(-12, 12, size=(n, d)))).astype(np.float32),\n"
        "             'cancel': np.concatenate([base[:, :d // 2], -base[:, :d - d // 2] * (1 + 1e-3 * rng.standard_normal((n, d - d // 2)))], axis=1).astype(np.float32)}\n"
        "    for name, g in cases.items():\n"
        "        for metric in (0, 1):\n"
        "            gg = ro.canon_normalize_rows(g) if (metric == 0 and name != 'dynamic') else np.ascontiguousarray(g)\n"
        "            qs = np.ascontiguousarray(gg[rng.permutation(n)[:q]])\n"
        "            dt, eps = probe(qs, gg, metric)\n"
        "            exact = ro.canon_pdist(qs, gg, metric)                      # [q, n]: the canonical fp32 chain\n"
        "            err = np.abs(dt.T.astype(np.float64) - exact.astype(np.float64))\n"
        "            assert np.isfinite(eps).all() and (eps > 0).all()\n"
        "            ratio = float((err / eps[:, None].astype(np.float64)).max())\n"
        "            assert ratio <= 1.0, (d, name, metric, ratio)\n"
        "            worst_bound = max(worst_bound, ratio)\n"
        "            if metric == 0:\n"
        "                a, b = image(qs), image(gg)\n"
        "                v64 = a @ b.T; s64 = np.abs(a) @ np.abs(b).T\n"
        "                kp = (d + 127) // 128 * 128\n"
        "                a1 = np.abs(-dt.T.astype(np.float64) - v64) / ((kp / 16) * 2.0 ** -20 * s64 + 1e-300)\n"
        "                worst_a1 = max(worst_a1, float(a1.max()))\n"
        "print('worst |d~ - d| / eps = %.4f; worst accumulation error / A1 allowance = %.5f' % (worst_bound, worst_a1))\n"
        "assert worst_a1 <= 0.125\n")
    print([ln for ln in out.splitlines() if ln.startswith("worst")])


def test_prefilter_adversarial_near_duplicates_around_rank_k():
    """A gallery built to defeat a filter that trusted its bf16 scores: hundreds of near-duplicates of every query whose distances
    differ by single float32 ulps around rank k -- far below eps -- plus exact duplicates (tie groups straddling rank k).  The
    refinement must recompute the whole eps window exactly and still return the canonical head, bit for bit; product library, no
    switches, both metrics, K-blocks at D = 1000."""
    import sehip as m
    rng = np.random.default_rng(21)
    for d, kb, metric in ((100, None, 0), (100, None, 1), (1000, [448, 276, 276], 0)):
        n, q, k = 17000, 48, 251
        g = rng.standard_normal((n, d)).astype(np.float32)
        if metric == 0:
            g = ro.canon_normalize_rows(g)
        qs = np.ascontiguousarray(g[:q])
        # rows 1000 .. 1000 + 48 * 330: 330 perturbed copies of each query, the perturbation a few float32 ulps of one coordinate
        for i in range(q):
            blk = g[1000 + i * 330: 1000 + (i + 1) * 330]
            blk[:] = qs[i]
            cols = rng.integers(0, d, size=330)
            ulps = rng.integers(-6, 7, size=330)
            v = (blk[np.arange(330), cols].view(np.int32) + ulps).astype(np.int32)
            blk[np.arange(330), cols] = v.view(np.float32)
            blk[::11] = qs[i]                                   # exact duplicates: ties broken by index
        dd, ii = m.retrieve_topk(dev(qs), dev(g), k, metric=metric, kblocks=kb)
        wd, wi = want_topk(qs, g, k, metric, kb)
        assert np.array_equal(ii.cpu().numpy(), wi), (d, metric)
        assert np.array_equal(dd.cpu().numpy(), wd), (d, metric)


def test_prefilter_irregular_rows():
    """Rows the bf16 bound says nothing about -- NaN / +-inf entries, magnitudes >= 2^60, all-zero rows, denormals -- in the gallery and
    among the queries: their bf16 images are NaN, every d~ with them is NaN, NaN always becomes a candidate and the exact chain decides
    (irregular queries are redone over the whole gallery).  Same bits as the oracle."""
    run_with_tuning_lib(
        "os.environ['SE_TOPK_FUSED'] = '1'\n"
        "rng = np.random.default_rng(33)\n"
        "n, d, k = 5000, 72, 60\n"
        "g = rng.standard_normal((n, d)).astype(np.float32)\n"
        "g[17, 3] = np.inf; g[18, 5] = -np.inf; g[19, 0] = np.nan; g[20] = 0.0; g[21, 7] = 3e30; g[22, 1] = -2e19; g[23] *= 1e-41\n"
        "g[24, 3] = np.inf; g[24, 4] = -np.inf\n"
        "qs = g[[0, 1, 17, 18, 19, 20, 21, 22, 23, 24, 4999]].copy()\n"
        "qs[0, 2] = 1e-42\n"
        "for metric in (0, 1):\n"
        "    with np.errstate(all='ignore'):\n"
        "        wd, wi = T.want_topk(qs, g, k, metric)\n"
        "    dd, ii = sehip.retrieve_topk(torch.from_numpy(qs).cuda(), torch.from_numpy(g).cuda(), k, metric=metric)\n"
        "    assert np.array_equal(ii.cpu().numpy(), wi), metric\n"
        "    assert np.array_equal(dd.cpu().numpy(), wd, equal_nan=True), metric\n")


def test_prefilter_statistics_at_full_size():
    """What the pre-filter does at BASELINE configs[2] size (tuning build, SE_TOPK_VERBOSE): candidates per query the bf16 pass
    admits, exact recomputations per query, queries sent to the exact fallback (< 1e-3 of them)."""
    out = run_with_tuning_lib(
        "os.environ['SE_TOPK_VERBOSE'] = '1'\n"
        "x = torch.from_numpy(np.random.default_rng(0).standard_normal((50000, 100)).astype(np.float32)).cuda()\n"
        "sehip.normalize_rows_(x)\n"
        "d, i = sehip.retrieve_topk(x, x, 251, metric=0)\n"
        "assert bool((i[:, 0] == torch.arange(50000, device='cuda', dtype=torch.int32)).all())\n")
    line = [ln for ln in out.splitlines() if "prefilter:" in ln][-1]
    print(line)
    redo = int(line.split("redo=")[1].split()[0])
    assert redo <= 50, line


@pytest.mark.parametrize("side_work", ["pdist", "topk"])
def test_topk_rows_stays_sorted_while_a_second_stream_is_busy(sehip, side_work):
    """Regression (round 4): with another stream's kernels sharing the CUs, se_topk_rows (k = 251 -> a 256-entry LDS bitonic
    sort over two waves) returned the right entries in the wrong order in ~1 of 5 calls: one s_barrier of the sort loop was
    compiled without the LDS wait in front of it.  tools/stress_one_proc.py is the long version of this test."""
    rng = np.random.default_rng(0)
    gallery = rng.standard_normal((1501, 200)).astype(np.float32)
    gh = ro.canon_normalize_rows(gallery)
    pd = ro.canon_pdist(gh[:300], gh, 0)
    k = 251
    wd, wi = ro.canon_topk_rows(pd, k)
    pdg = dev(pd)
    big = dev(rng.standard_normal((6000, 200)).astype(np.float32))
    side = torch.cuda.Stream()
    wrong = 0
    for it in range(120):
        with torch.cuda.stream(side):
            for _ in range(3):
                if side_work == "pdist":
                    sehip.pairwise_dist(big, big, metric=0)
                else:
                    sehip.topk_rows(pdg, k)
        d, i = sehip.topk_rows(pdg, k)
        wrong += not (np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), wd))
    torch.cuda.synchronize()
    assert wrong == 0, "%d of 120 calls differ from the canonical ranking" % wrong
