"""GPU parity tests for the retrieval hot path: HIP kernels (through the C ABI) vs the oracle.

Bar: bit-exact distances under the canonical arithmetic and bit-exact ranking indices
(BASELINE.json north_star: "bit-exact for retrieval ranking indices").
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as ro

pytestmark = pytest.mark.gpu

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT_DIR, 'semantic-embeddings_amd')


@pytest.fixture(scope="module")
def sehip():
    import sehip as m
    m.lib()
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gauss(n, d, seed=0):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


# ---------------------------------------------------------------- row norms (NumPy pairwise order)

@pytest.mark.parametrize("d", [1, 5, 7, 8, 9, 15, 16, 17, 63, 64, 100, 127, 128, 129, 136, 143, 200, 247, 248, 249, 250, 255, 256, 257, 555, 1000, 2048, 2500])
def test_row_sqnorm_and_normalize_bit_exact(sehip, d):
    """D < 256: one lane per row; 256 <= D <= 4096: one wave per row (leaves of NumPy's pairwise tree dealt to 8-lane groups)."""
    x = gauss(131, d, seed=d)
    sq = sehip.row_sqnorm(dev(x)).cpu().numpy()
    assert np.array_equal(sq, np.sum(x ** 2, axis=-1))          # NumPy itself
    assert np.array_equal(sq, ro.canon_row_sqsum(x))            # C restatement
    xn = sehip.normalize_rows_(dev(x)).cpu().numpy()
    ref = x.copy()
    ref /= np.linalg.norm(ref, axis=-1, keepdims=True)
    assert np.array_equal(xn, ref)


@pytest.mark.parametrize("d", [260, 263, 300, 511, 1023, 1024, 1025, 1999, 3001, 4095, 4096, 4097, 5000])
def test_row_norms_wave_per_row_kernel_shapes(sehip, d):
    """The wave-per-row kernel on both sides of its range (256 ... 4096), odd widths (element loads instead of 16-byte ones),
    leaf tails (D % 8 != 0), unaligned row pitches (a column slice of a wider matrix), more rows than resident waves, and rows whose
    squares span 40 binades (the summation order is what decides the bits)."""
    rng = np.random.default_rng(d)
    for rows, ld in ((131, d), (37, d + 3), (9000 if d <= 1025 else 600, d + 4)):
        wide = (rng.standard_normal((rows, ld)) * np.exp2(rng.integers(-20, 20, size=(rows, ld)))).astype(np.float32)
        x = wide[:, :d]
        xd = dev(wide)[:, :d]
        assert xd.stride(0) == ld
        sq = sehip.row_sqnorm(xd).cpu().numpy()
        assert np.array_equal(sq, np.sum(np.ascontiguousarray(x) ** 2, axis=-1))
        assert np.array_equal(sq, ro.canon_row_sqsum(np.ascontiguousarray(x)))
        xn = sehip.normalize_rows_(xd).cpu().numpy()
        ref = np.ascontiguousarray(x).copy()
        ref /= np.linalg.norm(ref, axis=-1, keepdims=True)
        assert np.array_equal(xn, ref)


def test_normalize_zero_row_gives_nan_like_numpy(sehip):
    x = gauss(70, 16)
    x[3] = 0
    xn = sehip.normalize_rows_(dev(x)).cpu().numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = x / np.linalg.norm(x, axis=-1, keepdims=True)
    assert np.array_equal(np.isnan(xn), np.isnan(ref))
    assert np.array_equal(xn[~np.isnan(ref)], ref[~np.isnan(ref)])


# ---------------------------------------------------------------- distance kernel

@pytest.mark.parametrize("metric", [ro.METRIC_COSINE, ro.METRIC_EUCLID, ro.METRIC_DOT])
@pytest.mark.parametrize("q,n,d", [(1, 1, 1), (3, 5, 2), (37, 129, 7), (128, 128, 64), (130, 257, 100),
                                   (200, 300, 101), (65, 400, 200), (300, 70, 448)])
def test_pairwise_dist_bit_exact(sehip, metric, q, n, d):
    a = gauss(q, d, seed=1)
    b = gauss(n, d, seed=2)
    got = sehip.pairwise_dist(dev(a), dev(b), metric=metric).cpu().numpy()
    want = ro.canon_pdist(a, b, metric)
    assert np.array_equal(got, want)


def test_pairwise_dist_self_is_symmetric_and_transpose_detecting(sehip):
    x = gauss(260, 100, seed=5)
    got = sehip.pairwise_dist(dev(x), None, metric=ro.METRIC_COSINE).cpu().numpy()
    assert np.array_equal(got, got.T)
    assert np.array_equal(got, ro.canon_pdist(x, None, ro.METRIC_COSINE))
    # asymmetric operands: a transposed write would be caught here
    a = gauss(64, 32, seed=7)
    b = np.arange(96 * 32, dtype=np.float32).reshape(96, 32) / 100
    got = sehip.pairwise_dist(dev(a), dev(b), metric=ro.METRIC_DOT).cpu().numpy()
    assert np.array_equal(got, ro.canon_pdist(a, b, ro.METRIC_DOT))


@pytest.mark.parametrize("d,kblocks", [(555, [278, 277]), (1000, [448, 276, 276]), (130, [64, 66]), (100, [100])])
def test_pairwise_dist_kblocks(sehip, d, kblocks):
    a = gauss(70, d, seed=3)
    b = gauss(140, d, seed=4)
    got = sehip.pairwise_dist(dev(a), dev(b), metric=ro.METRIC_COSINE, kblocks=kblocks).cpu().numpy()
    assert np.array_equal(got, ro.canon_pdist(a, b, ro.METRIC_COSINE, kblocks))


def test_pairwise_dist_strided_rows(sehip):
    big = dev(gauss(100, 128, seed=9))
    a = big[:, :100]                       # ld = 128, d = 100
    got = sehip.pairwise_dist(a, None, metric=ro.METRIC_DOT).cpu().numpy()
    assert np.array_equal(got, ro.canon_pdist(a.cpu().numpy(), None, ro.METRIC_DOT))
    odd = dev(gauss(50, 103, seed=10))[:, 1:101]   # misaligned rows -> scalar staging path
    got = sehip.pairwise_dist(odd, None, metric=ro.METRIC_DOT).cpu().numpy()
    assert np.array_equal(got, ro.canon_pdist(odd.cpu().numpy(), None, ro.METRIC_DOT))


# ---------------------------------------------------------------- ranking

def tie_heavy(q, n, seed=0):
    rng = np.random.default_rng(seed)
    pd = rng.integers(-3, 4, size=(q, n)).astype(np.float32)     # lots of exact ties
    pd[0, :5] = [0.0, -0.0, np.nan, np.inf, -np.inf]
    pd[1, ::7] = np.nan
    return pd


@pytest.mark.parametrize("q,n", [(1, 1), (2, 63), (5, 64), (7, 65), (33, 1000), (9, 4097), (3, 50000)])
def test_rank_rows_matches_canonical_order(sehip, q, n):
    pd = gauss(q, n, seed=n)
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))
    assert np.array_equal(got, np.argsort(pd, axis=-1, kind="stable"))


def test_rank_rows_ties_nan_negzero_int64(sehip):
    pd = tie_heavy(40, 700)
    want = ro.canon_rank_rows(pd)
    got32 = sehip.rank_rows(dev(pd)).cpu().numpy()
    got64 = sehip.rank_rows(dev(pd), idx64=True).cpu().numpy()
    assert got64.dtype == np.int64
    assert np.array_equal(got32, want)
    assert np.array_equal(got64, want)


@pytest.mark.parametrize("n", [1023, 1025, 4096, 4097, 6144, 6145, 10000, 10241, 15360, 15361, 20481, 23552, 23553, 26624, 26625, 29696, 29697, 32768, 32769,
                               36864, 36865, 40961, 45056, 45057, 50000, 50177, 53248, 53249, 65536, 70001, 100000])
def test_rank_rows_register_kernel_boundaries(sehip, n):
    """Every instantiation of the register-resident kernel (2 / 8 / 12 / 20 / 30 / 40 / 46 / 52 / 58 / 64 / 72 / 80 / 88 / 98 / 104 keys per thread), its last
    full / first ragged step, and the hand-over to the sorted-runs path above 53248 columns; rows mix
    gaussian keys with exact-tie runs, NaN, infinities and signed zeros."""
    rng = np.random.default_rng(n)
    pd = rng.standard_normal((3, n)).astype(np.float32)
    pd[1] = rng.integers(-2, 3, size=n).astype(np.float32)          # five distinct values: long tie runs
    pd[2, ::3] = pd[2, 0]
    pd[2, 1:min(n, 9)] = np.array([np.nan, 0.0, -0.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38], dtype=np.float32)[:max(0, min(n, 9) - 1)]
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))


@pytest.mark.parametrize("n", [700, 4097, 50000])
def test_rank_rows_skewed_top_digit_rows(sehip, n):
    """All-positive (Euclidean-like) rows put every key on one counter of the most significant digit: the skew
    detector routes the call to the group-peeling kernel; rows it sampled vs rows it did not may differ in sign mix."""
    rng = np.random.default_rng(n)
    pd = (200.0 + 20.0 * rng.standard_normal((7, n))).astype(np.float32)     # detector rows 0, Q/2, Q-1: skewed
    pd[1] = rng.standard_normal(n).astype(np.float32)                          # a mixed-sign row inside a "skewed" call
    pd[2, ::5] = pd[2, 0]                                                      # ties
    pd[4] = 3.0                                                                # one value
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))
    pd2 = rng.standard_normal((5, n)).astype(np.float32)                      # detector says "not skewed" ...
    pd2[2] = np.abs(pd2[2]) + 100.0                                            # ... but one row is
    got2 = sehip.rank_rows(dev(pd2)).cpu().numpy()
    assert np.array_equal(got2, ro.canon_rank_rows(pd2))


@pytest.mark.parametrize("n", [32768, 40961, 50000, 53248])
def test_rank_rows_two_pass_path(sehip, n):
    """Long rows whose keys (all but a few) lie within 2^24 codes of the row maximum -- Euclidean-distance rows: the query's own
    distance is the outlier -- take the two-pass path of the register-resident kernel (detector flag 2); every row re-checks itself:
    up to 256 keys below the window are put in order afterwards (here: none, 1, exactly 256, with ties / zeros / negatives among
    them, and at columns the detector does not sample), 257 send the row back to the three passes, as do NaN-free rows that are too
    wide.  Also exact ties inside the window, keys exactly on the window's lower edge, +inf (window anchored at inf: everything else
    is 'below'), NaN and padding-like all-ones."""
    rng = np.random.default_rng(n)
    q = 13
    pd = (200.0 + 25.0 * rng.standard_normal((q, n))).astype(np.float32).clip(120.0, 300.0)
    free = np.setdiff1d(np.arange(n), (np.arange(1024) * n) // 1024)         # columns the detector's 1024-column sample skips
    pd[0, 0] = 0.0                                                           # the query's own distance (sampled: one below per row is allowed)
    pd[1, free[:256]] = rng.choice(np.array([0.0, -0.0, 1e-3, -1e-3, 2.5, 2.5, -7.0], dtype=np.float32), size=256)   # 256 below, with ties
    pd[2, free[:257]] = np.linspace(-1.0, 1.0, 257, dtype=np.float32)        # 257 below: three passes for this row
    pd[3] = np.round(pd[3])                                                  # ~180 distinct values: long tie runs inside the window
    pd[4] = 210.0                                                            # one value
    m = np.float32(pd[5].max())
    edge = (m.view(np.uint32) - np.uint32((1 << 24) - 3)).view(np.float32)   # the smallest key still inside the window of row 5
    pd[5, free[:6]] = np.array([edge, np.nextafter(edge, np.float32(0)), edge, np.nextafter(edge, np.float32(1e9)), 0.5, edge], dtype=np.float32)
    pd[6, free[10]] = np.inf                                                 # window at +inf: the row falls back
    pd[7, free[3:9]] = np.nan                                                # NaN keys sort last on either path
    pd[8] = (1e-3 * np.abs(rng.standard_normal(n)) + 1e-6).astype(np.float32)   # positive but 2^30 codes wide: three passes
    pd[9, free[:20]] = -np.abs(rng.standard_normal(20)).astype(np.float32)   # 20 distinct negatives below the window
    # NaNs with the SIGN BIT set (0xFFC00000 is what 0 / 0 gives on x86, FMA chains propagate it): negative as raw integers, yet they
    # sort last like every NaN (round-5 advisor finding: the raw-key window test ranked them first)
    neg_nan = np.array([0xFFC00000, 0xFFFFFFFF, 0xFF800001], dtype=np.uint32).view(np.float32)
    pd[10, free[40:43]] = neg_nan
    pd[11, free[5]] = neg_nan[0]
    pd[11, free[6]] = np.nan
    pd[11, 0] = 0.0
    pd[12, free[:300:2]] = neg_nan[0]                                        # 150 of them: more than a few, fewer than the outlier cap
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    want = ro.canon_rank_rows(pd)
    for r in range(q):
        assert np.array_equal(got[r], want[r]), r
    got64 = sehip.rank_rows(dev(pd[:3]), idx64=True).cpu().numpy()           # the reference's index dtype through the same path
    assert np.array_equal(got64, want[:3])


def image_path_rows(rng, n):
    """Rows aimed at the image path of the register-resident kernel (detector flag 3: two passes on a 24-bit image of the key + repair
    of the keys that share an image): collisions of every run length, runs whose sorted positions straddle the scan's group
    boundaries, rows the image cannot take (NaN, infinities, all zero, tiny), rows that overflow the worklist or the run cap (sorted
    again with three passes), exponent-choice boundaries."""
    rows = []
    rows.append((0.1 * rng.standard_normal(n)).astype(np.float32))                       # 0: cosine-like
    v = (0.1 * rng.standard_normal(n)).astype(np.float32)                                 # 1: self distance just below -1, exact duplicates
    v[7] = np.float32(-1.0000001)
    v[n // 2] = v[3]
    v[n - 1] = v[3]
    rows.append(v)
    g = (rng.integers(0, 1 << 21, size=n) - (1 << 20)).astype(np.float32) * np.float32(2.0 ** -24)   # 2: grid near zero under a magnitude-1 key
    g[0] = -1.0
    rows.append(g)
    g = (rng.integers(0, 1 << 15, size=n) - (1 << 14)).astype(np.float32) * np.float32(2.0 ** -24)   # 3: dense grid: worklist overflow / long runs
    g[5] = 1.0
    rows.append(g)
    rows.append(np.exp(rng.uniform(-3, 3, size=n)).astype(np.float32))                   # 4: all positive, wide
    rows.append(rng.choice(np.array([0.5, -0.25, 0.125, 0.7], dtype=np.float32), size=n))   # 5: four values: every image a long run
    z = (1e-3 * rng.standard_normal(n)).astype(np.float32)                                # 6: zeros of both signs among small values
    z[::7] = 0.0
    z[3::7] = -0.0
    rows.append(z)
    w = (0.1 * rng.standard_normal(n)).astype(np.float32)                                 # 7: NaN / inf present: not an image row
    w[11] = np.nan
    w[n - 3] = np.inf
    rows.append(w)
    rows.append(np.zeros(n, dtype=np.float32))                                            # 8: all zero
    base = np.sort((0.3 * rng.random(n) + 0.1).astype(np.float32))                        # 9: runs of 2 .. 9 keys inside one image cell
    v = base.copy()                                                                       #    (positive: image ulp 2^-22 for c = 2), every 109 sorted positions
    for start in range(100, n - 20, 109):
        L = 2 + (start // 109) % 8
        lo = np.float32(np.floor(base[start] * 2 ** 22) / 2 ** 22)
        v[start:start + L] = lo + np.arange(L, 0, -1).astype(np.float32) * np.float32(2.0 ** -25)   # descending by index: the repair reverses them
    v[0] = -1.0
    rows.append(v)
    rows.append(v[rng.permutation(n)])                                                    # 10: the same keys in random columns
    a = (0.4 * rng.standard_normal(n)).astype(np.float32)                                 # 11 / 12: largest magnitude exactly 1.5 / just above
    a[1] = 1.5
    rows.append(a)
    b = a.copy()
    b[1] = np.float32(1.5000001)
    rows.append(b)
    rows.append((1e-35 * rng.standard_normal(n)).astype(np.float32))                     # 13: below 2^-100: three passes
    rows.append((1e37 * rng.standard_normal(n)).astype(np.float32))                      # 14: huge magnitudes
    rows.append((-200.0 + 20.0 * rng.standard_normal(n)).astype(np.float32))             # 15: negative, narrow
    c = (0.05 * rng.standard_normal(n)).astype(np.float32)                                # 16: clustered: 40 centres + 1e-6 noise
    c = (rng.choice(c[:40], size=n) + 1e-6 * rng.standard_normal(n)).astype(np.float32)
    rows.append(c)
    d = (0.1 * rng.standard_normal(n)).astype(np.float32)                                 # 17: every key 2 .. 6 times (tie runs in index order)
    d = np.repeat(d[: n // 2], 6)[:n][rng.permutation(n)]
    rows.append(d)
    return np.stack(rows)


@pytest.mark.parametrize("n", [32768, 36000, 40961, 45000, 50000, 50176])
def test_rank_rows_image_path(sehip, n):
    """Cosine-like calls (mixed signs, no dominant most significant digit: detector flag 3) of the long-row instantiations."""
    rng = np.random.default_rng(n)
    pd = image_path_rows(rng, n)
    pd = np.concatenate([pd, pd[::-1]], axis=0)        # (a workgroup's back-off state sees fit and unfit rows in both orders)
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    want = ro.canon_rank_rows(pd)
    for r in range(pd.shape[0]):
        assert np.array_equal(got[r], want[r]), r
    got64 = sehip.rank_rows(dev(pd[:4]), idx64=True).cpu().numpy()
    assert np.array_equal(got64, want[:4])


def test_rank_rows_image_path_many_rows_per_workgroup(sehip):
    """600 cosine rows at 50,000 columns with unfit rows mixed in: every workgroup sorts several rows, gives some up and backs off."""
    rng = np.random.default_rng(77)
    n = 50000
    x = rng.standard_normal((n, 64)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    pd = np.ascontiguousarray(-(x[:600] @ x.T))
    pd[5::97] = rng.choice(np.array([0.5, -0.25], dtype=np.float32), size=(len(pd[5::97]), n))
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))


def test_rank_rows_detector_picks_the_variant_in_subprocess():
    """Which build the skew detector selects (the phase profile of the tuning build names the variant that ran): the reference's cosine
    rows take the image path (3), its Euclidean rows the two lossless passes (2), four-valued rows (no dominant digit, but nothing an
    image could separate) the plain three passes (0), two-valued rows the group-peeling build (1)."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "rng = np.random.default_rng(1)\n"
        "n = 50000\n"
        "for name, pd in (('cos', 0.1 * rng.standard_normal((300, n))), ('euc', 200.0 + 20.0 * rng.standard_normal((300, n))),\n"
        "                 ('few', rng.choice(np.array([1.0, 2.0, 3.0, 4.0]), size=(300, n))),\n"
        "                 ('two', rng.choice(np.array([1.0, 2.0]), size=(300, n)))):\n"
        "    print('CASE', name, file=sys.stderr, flush=True)\n"
        "    sehip.rank_rows(torch.from_numpy(pd.astype(np.float32)).cuda())\n"
        "    torch.cuda.synchronize()\n"
    ) % ([PKG_DIR, ROOT_DIR],)
    env = dict(os.environ, SE_RR_PROFILE="1", SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    seen = {}
    case = None
    for line in out.stdout.splitlines():
        if line.startswith("CASE"):
            case = line.split()[1]
        elif "[se_rank_rows profile] ITEMS=98" in line and case:
            seen.setdefault(case, set()).add(line.split("peel=")[1].split()[0])
    assert seen == {"cos": {"3"}, "euc": {"2"}, "few": {"0"}, "two": {"1"}}, (seen, out.stdout[-2000:])


@pytest.mark.parametrize("n", [50000, 40961, 33000, 3001, 70])
def test_rank_rows_uint16_output(sehip, n):
    """idx64 == 2 of se_rank_rows: uint16 ranks (int16 tensors hold the bit patterns -- indices above 32,767 read as negative int16)
    from every variant of the register-resident kernel == the int32 ranks; the order guard reads them; rows above 53,248 columns are
    refused (SE_ERR_UNSUPPORTED), not silently truncated."""
    rng = np.random.default_rng(n)
    rows = [(0.1 * rng.standard_normal(n)).astype(np.float32),                     # cosine-like: image path at long rows
            (200.0 + 20.0 * rng.standard_normal(n)).astype(np.float32),            # Euclid-like: window path
            rng.choice(np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32), size=n),  # few values: plain three passes
            np.zeros(n, dtype=np.float32)]
    for base in rows:
        pd = np.stack([base, base[::-1].copy(), np.roll(base, 17)])
        d = dev(pd)
        r32 = sehip.rank_rows(d)
        r16 = sehip.rank_rows(d, idx16=True)
        assert r16.dtype == torch.int16
        assert np.array_equal(r16.cpu().numpy().view(np.uint16).astype(np.int32), r32.cpu().numpy())
        assert sehip.rank_rows_check(d, r16) == 0
    # unaligned / strided output: element stores
    out = torch.empty((3, n + 3), dtype=torch.int16, device="cuda")[:, 1:n + 1]
    sehip.rank_rows(d, out=out)
    assert np.array_equal(out.cpu().numpy().view(np.uint16).astype(np.int32), r32.cpu().numpy())
    with pytest.raises(sehip.SehipError):
        sehip.rank_rows(torch.zeros((2, 53249), device="cuda"), idx16=True)


def test_rank_rows_strided_and_unaligned_output(sehip):
    """Row pitches that are not multiples of 16 bytes (scalar write-out) and a strided input."""
    pdw = gauss(6, 3001, seed=5)
    pd = dev(pdw)[:, 1:2998]                                       # ld 3001, first element misaligned
    out = torch.empty((6, 2999), dtype=torch.int32, device="cuda")[:, 1:2998]
    sehip.rank_rows(pd, out=out)
    assert np.array_equal(out.cpu().numpy(), ro.canon_rank_rows(np.ascontiguousarray(pdw[:, 1:2998])))
    out64 = sehip.rank_rows(pd, idx64=True)
    assert np.array_equal(out64.cpu().numpy(), ro.canon_rank_rows(np.ascontiguousarray(pdw[:, 1:2998])))


def test_rank_rows_ballot_kernel_in_subprocess():
    """The guaranteed-order (ballot multisplit) variant stays covered although the capability probe selects
    the hardware-ordered variant on MI355X: SE_RANK_SAFE is read once per process, hence the subprocess."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "rng = np.random.default_rng(3)\n"
        "for n in (777, 5000, 20481, 50000):\n"
        "    pd = rng.standard_normal((3, n)).astype(np.float32)\n"
        "    pd[1] = rng.integers(-2, 3, size=n).astype(np.float32)\n"
        "    pd[2, ::3] = np.nan\n"
        "    got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()\n"
        "    assert np.array_equal(got, ro.canon_rank_rows(pd)), n\n"
        "print('ballot-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR],)
    env = dict(os.environ, SE_RANK_SAFE="1", SE_RANK_VERBOSE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "ballot-ok" in out.stdout, out.stdout
    assert "hardware-ordered" not in out.stdout


@pytest.mark.parametrize("peel", ["0", "1", "2", "3"])
def test_rank_rows_pinned_peel_variants_in_subprocess(peel):
    """All builds of the hardware-ordered kernel (plain / group-peeling last pass / two-pass path / image path, the last two for long rows only) on
    every row shape, whatever the skew detector would choose: SE_RANK_PEEL pins the build (read once per process, hence the subprocess).  Long rows use
    the 12-bit last digit whose counters alias the exchange buffer; all-positive rows make lane 0's digit group large."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "rng = np.random.default_rng(5)\n"
        "for n in (700, 5000, 20481, 32768, 40961, 50000, 53248):\n"
        "    pd = rng.standard_normal((4, n)).astype(np.float32)\n"
        "    pd[1] = 150.0 + 30.0 * np.abs(pd[1])\n"                     # Euclidean-like: one exponent, all positive
        "    pd[2] = rng.integers(0, 3, size=n).astype(np.float32)\n"    # three values: huge tie groups
        "    pd[3, ::7] = np.nan\n"
        "    got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()\n"
        "    assert np.array_equal(got, ro.canon_rank_rows(pd)), n\n"
        "print('peel-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR],)
    # variant switches exist only in the -DSE_TUNING build of the library (the product ignores them)
    env = dict(os.environ, SE_RANK_PEEL=peel, SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "peel-ok" in out.stdout, out.stdout


def test_rank_order_guard_accepts_rankings_and_counts_violations(sehip):
    """se_rank_rows_check: 0 for rank_rows' output (int32 and int64, ties / NaN / -0 included), the exact number of rows once
    adjacent ranks are swapped, out-of-range indices counted too."""
    rng = np.random.default_rng(12)
    pd = rng.standard_normal((40, 3000)).astype(np.float32)
    pd[3] = rng.integers(0, 4, size=3000).astype(np.float32)
    pd[4, ::5] = np.nan
    pd[5, ::3] = -0.0
    pd[5, 1::3] = 0.0
    pdd = dev(pd)
    for idx64 in (False, True):
        rk = sehip.rank_rows(pdd, idx64=idx64)
        assert sehip.rank_rows_check(pdd, rk) == 0
        bad = rk.clone()
        for row, r in ((0, 0), (7, 1499), (39, 2998)):
            bad[row, r], bad[row, r + 1] = rk[row, r + 1], rk[row, r]
        bad[20, 100] = 3000                                 # out of range
        assert sehip.rank_rows_check(pdd, bad) == 4
    # ties must be in index order: swapping two equal-distance neighbours is a violation as well
    rk = sehip.rank_rows(pdd)
    keys = torch.gather(pdd[3:4], 1, rk[3:4].long())[0]
    r = int(torch.nonzero(keys[1:] == keys[:-1])[0])
    bad = rk.clone()
    bad[3, r], bad[3, r + 1] = rk[3, r + 1], rk[3, r]
    assert sehip.rank_rows_check(pdd, bad) == 1


@pytest.mark.parametrize("q", [40, 9000])
def test_rank_order_guard_repairs_injected_violations_in_subprocess(q):
    """SE_RANK_INJECT=1 (tuning build) swaps two adjacent ranks in every 7th row behind the hardware-ordered kernel -- what a lost
    stability would look like.  The guard behind the first ranking of the process must find those rows, re-rank them with the
    ballot kernel (row list: q = 40; more rows than the list holds -> whole call redone: q = 9000), say so, and leave the
    canonical ranking; the device then stays on the ballot kernel."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "sehip.ops._rank_ready.add(0)\n"      # no se_rank_rows_init: the LAZY first-call guard of se_rank_rows is what is being tested
        "rng = np.random.default_rng(9)\n"
        "pd = rng.standard_normal((%d, 2500)).astype(np.float32)\n"
        "got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()\n"
        "assert np.array_equal(got, ro.canon_rank_rows(pd))\n"
        "got = sehip.rank_rows(torch.from_numpy(pd[:50]).cuda(), idx64=True).cpu().numpy()\n"     # second call: ballot kernel, no injection left to repair
        "assert np.array_equal(got, ro.canon_rank_rows(pd[:50]))\n"
        "print('guard-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR], q)
    # SE_RANK_CHECK=1: every row is checked (the first-call guard of a process looks at 512 evenly spaced rows only)
    env = dict(os.environ, SE_RANK_INJECT="1", SE_RANK_CHECK="1", SE_RANK_VERBOSE="1", SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "guard-ok" in out.stdout, out.stdout
    assert "order guard" in out.stdout and "ballot kernel from now on" in out.stdout, out.stdout
    if q == 40:     # the same without the switch: the sampled first-call guard (all 40 rows are in its sample) finds the injected rows too
        env.pop("SE_RANK_CHECK")
        out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0 and "guard-ok" in out.stdout and "order guard" in out.stdout, out.stdout


def test_rank_rows_init_self_test_and_graph_capture():
    """se_rank_rows_init audits every hardware-ordered kernel variant on crafted tie-heavy rows (verdict lines under
    SE_RANK_VERBOSE=1, all clean on MI355X); afterwards se_rank_rows is purely asynchronous: it can be captured into a HIP graph
    (a synchronising call would fail the capture) and the replayed graph re-ranks new distances in place -- bit-equal to the oracle."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from sehip._lib import lib, ptr, check\n"
        "from oracle import retrieval_oracle as ro\n"
        "sehip.rank_rows_init()\n"
        "rng = np.random.default_rng(11)\n"
        "for n in (3000, 40000, 60000):\n"
        "    a = rng.standard_normal((6, n)).astype(np.float32); a[1] = rng.integers(-2, 3, size=n)\n"
        "    b = 100.0 + np.abs(rng.standard_normal((6, n))).astype(np.float32); b[2, ::4] = b[2, 0]\n"
        "    pd = torch.from_numpy(a).cuda(); rk = torch.empty((6, n), dtype=torch.int32, device='cuda')\n"
        "    ws = torch.empty((int(lib().se_rank_rows_workspace_bytes(6, n)),), dtype=torch.uint8, device='cuda')\n"
        "    side = torch.cuda.Stream()\n"
        "    side.wait_stream(torch.cuda.current_stream())\n"
        "    g = torch.cuda.CUDAGraph()\n"
        "    with torch.cuda.graph(g, stream=side):\n"
        "        check(lib().se_rank_rows(ptr(pd), pd.stride(0), 6, n, ptr(rk), 0, rk.stride(0), ptr(ws), ws.numel(),\n"
        "                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'se_rank_rows under capture')\n"
        "    g.replay(); torch.cuda.synchronize()\n"
        "    assert np.array_equal(rk.cpu().numpy(), ro.canon_rank_rows(a)), n\n"
        "    pd.copy_(torch.from_numpy(b)); g.replay(); torch.cuda.synchronize()\n"
        "    assert np.array_equal(rk.cpu().numpy(), ro.canon_rank_rows(b)), n\n"
        "print('init-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR],)
    env = dict(os.environ, SE_RANK_VERBOSE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "init-ok" in out.stdout, out.stdout[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("[se_rank_rows_init]")]
    assert len(lines) >= 5 and all(": 0 of 4 rows" in ln for ln in lines), out.stdout[-3000:]
    for what in ("short rows, plain", "short rows, group-peeling", "long rows, plain", "long rows, two-pass", "segment runs + merge"):
        assert any(what in ln for ln in lines), (what, lines)
    assert "order guard" not in out.stdout          # the lazy guard never ran: init had the verdict


def long_rows(n, seed):
    rng = np.random.default_rng(seed)
    pd = rng.standard_normal((6, n)).astype(np.float32)
    pd[1] = rng.integers(-2, 3, size=n).astype(np.float32)          # five values: every tie group spans both segments
    pd[2, ::3] = pd[2, 0]
    pd[2, 1:9] = np.array([np.nan, 0.0, -0.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38], dtype=np.float32)
    pd[3] = 150.0 + 30.0 * np.abs(pd[3])                            # Euclidean-like: one exponent
    pd[4, n // 2:] = pd[4, :n - n // 2]                             # the second half repeats the first: all ties go to the lower index
    pd[5, ::2] = np.nan                                             # NaN in both segments, sorted last in index order
    return pd


@pytest.mark.parametrize("n", [53249, 53256, 65536, 65537, 73728, 73729, 81920, 81921, 90112, 90113, 100352, 100353, 106496, 106497, 131073, 212992, 212993, 425984])
def test_rank_rows_long_rows_sorted_runs(sehip, n):
    """53,248 < N <= 425,984: 2 / 4 / 8 segments sorted by the register-resident kernel (all six long-row instantiations: 64 / 72 / 80 /
    88 / 98 / 104 keys per thread, first and last length of each) + merge tree (merge-path partition + tile merge per level; the levels
    before the last write (key, index) runs) == the canonical ranking, int32 and int64, ties across segment boundaries in index order."""
    pd = long_rows(n, n)
    want = ro.canon_rank_rows(pd)
    assert np.array_equal(sehip.rank_rows(dev(pd)).cpu().numpy(), want)
    assert np.array_equal(sehip.rank_rows(dev(pd), idx64=True).cpu().numpy(), want)
    assert sehip.lib().se_rank_rows_workspace_bytes(6, n) >= 6 * n * 6


def test_rank_rows_long_rows_chunks_strides_and_guard(sehip):
    """More rows than one chunk of the runs path (4,096), a strided unaligned input and output (scalar write-out of the merge),
    and the order guard's verdict on the result."""
    q, n = 4200, 53301
    x = torch.randn(q, n + 3, device="cuda")
    x[::2, ::5] = 0.25
    pd = x[:, 1:n + 1]
    out = torch.empty((q, n + 2), dtype=torch.int32, device="cuda")[:, 1:n + 1]
    sehip.rank_rows(pd, out=out)
    want = torch.argsort(pd, dim=1, stable=True)                    # finite keys without signed zeros: the canonical order
    assert bool((out.long() == want).all())
    assert sehip.rank_rows_check(pd, out) == 0
    out64 = sehip.rank_rows(pd[:300], idx64=True)
    assert bool((out64 == want[:300]).all())


def test_rank_rows_beyond_eight_segments_takes_the_tiled_kernel(sehip):
    pd = long_rows(425985, 3)[:3]
    assert np.array_equal(sehip.rank_rows(dev(pd)).cpu().numpy(), ro.canon_rank_rows(pd))


def test_rank_rows_pinned_tiled_kernel_in_subprocess():
    """SE_RANK_NORUNS=1 (tuning build) pins the tiled kernel for rows the merge tree would take: the fallback of the runs path (no
    hardware-order guarantee, or a guard violation) stays covered."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "sys.path.insert(0, %r)\n"
        "import test_gpu_retrieval as T\n"
        "for n in (60000, 106497, 131072):\n"
        "    pd = T.long_rows(n, 1)\n"
        "    got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()\n"
        "    assert np.array_equal(got, ro.canon_rank_rows(pd)), n\n"
        "print('tiled-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR], os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SE_RANK_NORUNS="1", SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "tiled-ok" in out.stdout, out.stdout


def test_rank_order_guard_behind_the_runs_path_in_subprocess():
    """SE_RANK_INJECT=1 behind the sorted-runs path: the first-call guard finds the swapped ranks, the call is redone by the tiled
    kernel and the device leaves the hardware-ordered paths (the next call, short rows, runs the ballot kernel)."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "import sehip\n"
        "from oracle import retrieval_oracle as ro\n"
        "sehip.ops._rank_ready.add(0)\n"      # no se_rank_rows_init: the lazy first-call guard is what is being tested
        "rng = np.random.default_rng(9)\n"
        "pd = rng.standard_normal((40, 60000)).astype(np.float32)\n"
        "got = sehip.rank_rows(torch.from_numpy(pd).cuda()).cpu().numpy()\n"
        "assert np.array_equal(got, ro.canon_rank_rows(pd))\n"
        "got = sehip.rank_rows(torch.from_numpy(pd[:, :3000]).cuda()).cpu().numpy()\n"
        "assert np.array_equal(got, ro.canon_rank_rows(pd[:, :3000]))\n"
        "print('guard-ok')\n"
    ) % ([PKG_DIR, ROOT_DIR],)
    env = dict(os.environ, SE_RANK_INJECT="1", SE_RANK_VERBOSE="1", SEHIP_LIB=os.path.join(PKG_DIR, "sehip", "libsehip_tuning.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "guard-ok" in out.stdout, out.stdout
    assert "order guard" in out.stdout and "tiled kernel" in out.stdout, out.stdout


@pytest.mark.parametrize("n", [300, 5000, 50000])
def test_rank_rows_special_values(sehip, n):
    """-0.0 ties with +0.0, denormals keep their order, infinities sit at the ends, every NaN (either sign, any payload) is last --
    all tie groups in index order.  (The register-resident kernel builds its keys with its own 6-instruction mapping.)"""
    rng = np.random.default_rng(n)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, np.inf, -np.inf, 3.4028235e38, -3.4028235e38], dtype=np.float32)
    pd = rng.choice(specials, size=(3, n)).astype(np.float32)
    pd[1] = np.where(rng.random(n) < 0.5, rng.standard_normal(n).astype(np.float32), pd[1])
    nanbits = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFFFFFFFF], dtype=np.uint32).view(np.float32)
    pd[2, ::5] = rng.choice(nanbits, size=len(pd[2, ::5]))
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))


def test_rank_rows_many_rows_persistent_grid(sehip):
    """More rows than resident workgroups: the persistent row loop re-uses LDS across rows."""
    pd = gauss(1500, 2500, seed=11)
    pd[::2, ::5] = 0.25
    got = sehip.rank_rows(dev(pd)).cpu().numpy()
    assert np.array_equal(got, ro.canon_rank_rows(pd))


@pytest.mark.parametrize("q,n,k", [(4, 10, 1), (4, 10, 10), (17, 300, 7), (9, 5000, 251), (3, 50000, 251), (5, 3000, 2048)])
def test_topk_rows_equals_head_of_full_ranking(sehip, q, n, k):
    pd = gauss(q, n, seed=k)
    d, i = sehip.topk_rows(dev(pd), k, col_offset=1000)
    wd, wi = ro.canon_topk_rows(pd, k, col_offset=1000)
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(d.cpu().numpy(), wd)


@pytest.mark.parametrize("n,k", [(8192, 1), (50000, 251), (50000, 2048), (20000, 100), (131072, 251)])
def test_topk_sample_select_and_repair_pass(sehip, n, k):
    """Long rows take the sample-select kernel; rows it cannot finish (all-equal row, a tie group of thousands
    straddling rank k, NaN-heavy row) are flagged and redone by the exact radix-select kernel behind it.  Every
    row must equal the head of the canonical full ranking."""
    rng = np.random.default_rng(n + k)
    pd = rng.standard_normal((6, n)).astype(np.float32)
    pd[1] = 0.5                                                      # one value: n candidates > capacity -> repair
    pd[2] = rng.integers(0, 3, size=n).astype(np.float32)            # three values: ties straddle rank k -> repair
    pd[3, ::2] = np.nan                                              # half NaN (sorted last)
    pd[4, : n // 2] = -np.abs(pd[4, : n // 2]) - 10.0                # low half far below the rest
    d, i = sehip.topk_rows(dev(pd), k, col_offset=7)
    wd, wi = ro.canon_topk_rows(pd, k, col_offset=7)
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(d.cpu().numpy(), wd, equal_nan=True)


def test_topk_rows_ties(sehip):
    pd = tie_heavy(30, 900, seed=3)
    pd[np.isnan(pd)] = 9.0
    for k in (1, 5, 129, 600, 900):
        d, i = sehip.topk_rows(dev(pd), k)
        wd, wi = ro.canon_topk_rows(pd, k)
        assert np.array_equal(i.cpu().numpy(), wi), k
        assert np.array_equal(d.cpu().numpy(), wd), k


def test_topk_merge_is_shard_invariant(sehip):
    # the sharded-gallery path: per-shard top-k + merge == top-k over the whole gallery
    q, n, d, k = 50, 1200, 100, 25
    x = gauss(n, d, seed=11)
    qs = x[:q]
    full = ro.canon_pdist(qs, x, ro.METRIC_COSINE)
    wd, wi = ro.canon_topk_rows(full, k)
    for parts in (2, 3, 8):
        bounds = np.linspace(0, n, parts + 1).astype(int)
        ds, is_ = [], []
        for p in range(parts):
            shard = x[bounds[p]:bounds[p + 1]]
            dd, ii = sehip.retrieve_topk(dev(qs), dev(shard), k, metric=ro.METRIC_COSINE, col_offset=int(bounds[p]))
            ds.append(dd); is_.append(ii)
        md, mi = sehip.topk_merge(torch.stack(ds), torch.stack(is_))
        assert np.array_equal(mi.cpu().numpy(), wi), parts
        assert np.array_equal(md.cpu().numpy(), wd), parts


# ---------------------------------------------------------------- end to end vs the reference's own output

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "retrieval_*.npz"))))
def test_golden_rankings_from_the_imported_reference(sehip, path):
    g = np.load(path)
    feats = g["features"].astype(np.float32)
    norm = bool(g["normalize"])
    ref = g["ref_ranking"].astype(np.int64)
    # D > 448: the reference's BLAS restarts its FMA chain per K block; the fixture carries the probed block list
    kblocks = g["kblocks"].tolist() if "kblocks" in g.files else None
    x = dev(feats.copy())
    if norm:
        sehip.normalize_rows_(x)
        pd = sehip.pairwise_dist(x, None, metric=ro.METRIC_COSINE, kblocks=kblocks)
    else:
        pd = sehip.pairwise_dist(x, None, metric=ro.METRIC_EUCLID, kblocks=kblocks)
    rk = sehip.rank_rows(pd).cpu().numpy().astype(np.int64)
    if "ids" in g.files:
        rk = g["ids"][rk]
    pdh = pd.cpu().numpy()
    # gate 1: exactly the canonical oracle
    cpd, crk = ro.canon_retrieval(feats, norm, kblocks=kblocks)
    assert np.array_equal(pdh, cpd)
    crk = crk.astype(np.int64)
    assert np.array_equal(rk, g["ids"][crk] if "ids" in g.files else crk)
    # gate 2: equal to the reference's (unstable-sort) output except inside exact-tie groups
    same = rk == ref
    if not same.all():
        pos = {int(v): i for i, v in enumerate(g["ids"])} if "ids" in g.files else None
        for r in np.nonzero(~same.all(axis=1))[0]:
            a = rk[r] if pos is None else np.array([pos[int(v)] for v in rk[r]])
            b = ref[r] if pos is None else np.array([pos[int(v)] for v in ref[r]])
            assert np.array_equal(pdh[r][a], pdh[r][b]), "row %d differs outside a tie group" % r


@pytest.mark.parametrize("branch", ["cos", "euc"])
def test_larger_golden_rankings_from_the_imported_reference(sehip, branch):
    """Round 5: the reference-ranking gate beyond 256 rows -- 4,096 clustered items with 64 exact duplicate rows, D = 100, both
    branches (tests/golden/bigretrieval_cluster.npz: the imported reference's rankings of 128 query rows).  Gate 1: distances and
    EVERY row's ranking equal the canonical oracle; gate 2: the sampled rows equal the reference except inside exact-tie groups."""
    g = np.load(os.path.join(ROOT_DIR, "tests", "golden", "bigretrieval_cluster.npz"))
    feats, rows = g["features"].astype(np.float32), g["rows"]
    x = dev(feats.copy())
    if branch == "cos":
        sehip.normalize_rows_(x)
        pd = sehip.pairwise_dist(x, None, metric=ro.METRIC_COSINE)
    else:
        pd = sehip.pairwise_dist(x, None, metric=ro.METRIC_EUCLID)
    rk = sehip.rank_rows(pd).cpu().numpy().astype(np.int64)
    pdh = pd.cpu().numpy()
    cpd, crk = ro.canon_retrieval(feats, branch == "cos")
    assert np.array_equal(pdh, cpd)
    assert np.array_equal(rk, crk.astype(np.int64))
    ref = g["ref_ranking_rows_" + branch].astype(np.int64)
    differing = 0
    for i, r in enumerate(rows):
        if not np.array_equal(rk[r], ref[i]):
            assert np.array_equal(pdh[r][rk[r]], pdh[r][ref[i]]), "row %d differs outside a tie group" % r
            differing += 1
    assert differing > 0     # the duplicates do force exact ties
    # the fused top-k head of the same problem
    d, i = sehip.retrieve_topk(x, x, 251, metric=ro.METRIC_COSINE if branch == "cos" else ro.METRIC_EUCLID,
                               sqq=None if branch == "cos" else sehip.row_sqnorm(x), sqg=None if branch == "cos" else sehip.row_sqnorm(x))
    assert np.array_equal(i.cpu().numpy().astype(np.int64), rk[:, :251])
    assert np.array_equal(d.cpu().numpy(), np.take_along_axis(pdh, rk[:, :251], axis=1))


def test_full_size_properties_50k(sehip):
    """BASELINE config 3 at full gallery size (50k x 100) on a 512-query tile: size-independent
    properties -- every row is a permutation, distances are sorted along the ranking, self is
    first for cosine, and the tile equals the canonical oracle on sampled rows."""
    n, d, q = 50000, 100, 512
    x = gauss(n, d, seed=0)
    xd = dev(x)
    sehip.normalize_rows_(xd)
    pd = sehip.pairwise_dist(xd[:q], xd, metric=ro.METRIC_COSINE)
    rk = sehip.rank_rows(pd)
    srt = torch.gather(pd, 1, rk.long())
    assert bool((srt[:, 1:] >= srt[:, :-1]).all())
    assert bool((rk.long().sort(dim=1).values == torch.arange(n, device="cuda")[None, :]).all())
    ties_ok = (srt[:, 1:] > srt[:, :-1]) | (rk[:, 1:] > rk[:, :-1])
    assert bool(ties_ok.all())
    xn = xd.cpu().numpy()
    rows = [0, 17, 511]
    want_pd = ro.canon_pdist(xn[rows], xn, ro.METRIC_COSINE)
    assert np.array_equal(pd[rows].cpu().numpy(), want_pd)
    assert np.array_equal(rk[rows].cpu().numpy(), ro.canon_rank_rows(want_pd))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "retrieval_d*[05]_*.npz"))))
def test_drop_in_reproduces_reference_beyond_448_with_kblocks(path):
    """`pairwise_retrieval(..., kblocks=...)` (the product path) on the D = 555 / D = 1000 fixtures == the imported
    reference's rankings outside exact-tie groups."""
    import evaluate_retrieval as er
    g = np.load(path)
    if "kblocks" not in g.files:
        pytest.skip("single-chain fixture")
    feats, norm = g["features"].astype(np.float32), bool(g["normalize"])
    ret = er.pairwise_retrieval(feats.copy(), normalize=norm, return_generator=False, kblocks=g["kblocks"].tolist())
    rk = np.array([ret[i] for i in range(len(feats))], dtype=np.int64)
    ref = g["ref_ranking"].astype(np.int64)
    cpd, _ = ro.canon_retrieval(feats, norm, kblocks=g["kblocks"].tolist())
    for r in np.nonzero((rk != ref).any(axis=1))[0]:
        assert np.array_equal(cpd[r][rk[r]], cpd[r][ref[r]]), "row %d differs outside a tie group" % r


@pytest.mark.parametrize("metric", ["cosine", "euclid"])
def test_benchmarked_step_at_full_size_is_oracle_exact(sehip, metric):
    """bench.py's exact step at BASELINE configs[2] size -- 50,000 x 50,000 x 100: normalise -> SYMMETRIC
    pairwise_dist(x, None) (upper-triangle tile walk + mirrored stores, 391 x 391 tiles) -> rank_rows (register-resident
    hardware-ordered kernel) -- checked by oracle/verify.py: matrix == its transpose bitwise, every row a permutation /
    sorted / index-ascending inside ties, and >= 50 sampled rows (first / last / middle tile rows, both sides of tile
    boundaries, random rows) bit-equal to canon.c's distances and canonical ranking."""
    from oracle import verify
    n, d = 50000, 100
    x = gauss(n, d, seed=0)
    g = dev(x)
    m = ro.METRIC_COSINE if metric == "cosine" else ro.METRIC_EUCLID
    if metric == "cosine":
        sehip.normalize_rows_(g)
        pd = sehip.pairwise_dist(g, None, metric=m)
    else:
        sq = sehip.row_sqnorm(g)
        pd = sehip.pairwise_dist(g, None, metric=m, sqa=sq, sqb=sq)
    rk = sehip.rank_rows(pd)
    ok, detail = verify.verify_retrieval_step(g.cpu().numpy(), pd, rk, m)
    assert ok, detail
    assert detail["rows_checked"] >= 50 and detail["symmetric"]
    # the reference's index dtype (int64) through the same kernels
    rk64 = sehip.rank_rows(pd[:1024], idx64=True)
    assert rk64.dtype == torch.int64 and bool((rk64 == rk[:1024].long()).all())
