"""HIP kernels (through the C ABI) against the oracle and the reference's golden vectors.
Run on the GPU box:  python -m pytest tests -m gpu
Tolerances: FP64 results rtol 1e-10 unless stated (different summation order / libm than NumPy);
integer / index results bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mcmc as omc  # noqa: E402
from oracle import philox as px  # noqa: E402
from oracle import ps  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def ctx_for(d, cap=0):
    from tempest_amd.device import HipContext
    return HipContext(d, 0, cap)


def soa(a, dev):
    """(n,d) host -> (d,n) device tensor"""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(dev)


def aos(t):
    return np.ascontiguousarray(t.cpu().numpy().T)


# ------------------------------------------------------------------------------------ RNG
def test_prior_draw_matches_philox(dev):
    for d in (1, 2, 5, 10):
        c = ctx_for(d)
        n = 1000
        u = c.empty(d, n)
        c.prior_draw(u, seed=0x1234567890ABCDEF, tick=7, item0=55)
        want = omc.prior_draw(n, d, 0x1234567890ABCDEF, 7, 55)
        np.testing.assert_array_equal(aos(u), want)
        assert want.min() >= 0 and want.max() < 1


# -------------------------------------------------------------------------- reweighting
def test_logmix_reweight_golden(dev):
    from tempest_amd.device import KEY_LOGMIX, KEY_LOGL
    g = load("g1_logw.npz")
    for k in range(int(g["n_cases"])):
        logl, bt, zt, nt = g[f"c{k}_logl"], g[f"c{k}_beta_t"], g[f"c{k}_logz_t"], g[f"c{k}_n_t"]
        c = ctx_for(3)
        c.history_load(None, None, logl, bt, zt, nt)
        assert c.size == logl.size and c.iterations == bt.size
        np.testing.assert_array_equal(c.history_read(KEY_LOGL), logl)
        cm = c.history_read(KEY_LOGMIX)
        np.testing.assert_allclose(cm, ps.log_mixture(logl, bt, zt, nt), rtol=1e-12, atol=1e-12, equal_nan=True)
        nh = int(nt.sum())
        for j in range(4):
            bf = float(g[f"c{k}_b{j}_beta"])
            ref_lw = g[f"c{k}_b{j}_logw_unnorm"]
            lw = c.logw(bf, nh).cpu().numpy()
            np.testing.assert_allclose(lw, ref_lw, rtol=1e-11, atol=1e-10, equal_nan=True)
            if not np.all(np.isfinite(ref_lw)):
                continue
            m, s1, s2 = c.reweight_eval([bf])[0]
            np.testing.assert_allclose(m + np.log(s1), g[f"c{k}_b{j}_logz"], rtol=1e-11, atol=1e-10)
            w_ref = np.exp(g[f"c{k}_b{j}_logw"])
            np.testing.assert_allclose(s1 * s1 / s2, ps.effective_sample_size(w_ref), rtol=1e-10)
            w = c.weights(bf, m, s1).cpu().numpy()
            np.testing.assert_allclose(w, w_ref, rtol=1e-9, atol=1e-300)
        c.close()


def test_nan_rows_propagate_like_reference(dev):
    # -inf log-likelihood at an iteration with beta_t = 0 gives 0*(-inf) = NaN in the reference
    g = load("g1_logw.npz")
    k = 3
    logl, bt, zt, nt = g[f"c{k}_logl"], g[f"c{k}_beta_t"], g[f"c{k}_logz_t"], g[f"c{k}_n_t"]
    assert np.isinf(logl).any()
    c = ctx_for(3)
    c.history_load(None, None, logl, bt, zt, nt)
    m, s1, s2 = c.reweight_eval([0.3])[0]
    assert np.isnan(g[f"c{k}_b2_logz"]) and np.isnan(m + np.log(s1))


def test_history_append_incremental(dev):
    from tempest_amd.device import KEY_LOGMIX, KEY_U, KEY_X
    rs = np.random.RandomState(0)
    d, n = 4, 777
    c = ctx_for(d)           # no capacity hint: exercises growth
    betas, logzs, us, xs, ls = [], [], [], [], []
    for t in range(9):
        u = rs.rand(n + t, d); x = 20 * u - 10
        l = -0.5 * np.sum(x ** 2, axis=1)
        beta, logz = (0.0 if t < 2 else 0.1 * t), -1.5 * t
        c.history_append(soa(u, dev), soa(x, dev), torch.from_numpy(l).to(dev), beta, logz)
        betas.append(beta); logzs.append(logz); us.append(u); xs.append(x); ls.append(l)
        nt = np.array([len(v) for v in ls])
        want = ps.log_mixture(np.concatenate(ls), betas, logzs, nt)
        np.testing.assert_allclose(c.history_read(KEY_LOGMIX), want, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(c.history_read(KEY_U), np.concatenate(us))
    np.testing.assert_array_equal(c.history_read(KEY_X), np.concatenate(xs))
    np.testing.assert_array_equal(c.history_read(KEY_X, 10, 5), np.concatenate(xs)[10:15])


@pytest.mark.parametrize("n", [1, 2, 3, 255, 4097, 1_000_003, 6_000_000])
def test_reweight_sizes_and_batching(dev, n):
    rs = np.random.RandomState(n % 1000)
    T = 16
    nt = np.full(T, n // T, dtype=np.int64); nt[-1] += n - nt.sum()
    nt = nt[nt > 0]; T = nt.size
    bt = np.linspace(0, 1, T) ** 2
    zt = -np.linspace(0, 30, T)
    logl = -rs.chisquare(10, size=n) * (1 + rs.rand(n))
    c = ctx_for(2)
    c.history_load(None, None, logl, bt, zt, nt)
    cm = ps.log_mixture(logl, bt, zt, nt)
    betas = np.array([0.0, 1e-4, 0.05, 0.37, 0.9, 1.0, 0.2, 0.21, 0.22, 0.23, 0.5, 0.51, 0.52, 0.53, 0.54])
    batch = c.reweight_eval(betas)
    for b, row in zip(betas[:4], batch[:4]):
        m, s1, s2 = ps.reweight_triple(logl, cm, b)
        np.testing.assert_allclose(row[0] + np.log(row[1]), m + np.log(s1), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(row[1] ** 2 / row[2], s1 ** 2 / s2, rtol=1e-10)
    for i, b in enumerate(betas):      # a batch is the same as one evaluation per beta
        one = c.reweight_eval([b])[0]
        np.testing.assert_allclose(one[0] + np.log(one[1]), batch[i][0] + np.log(batch[i][1]), rtol=1e-13)
        np.testing.assert_allclose(one[1] ** 2 / one[2], batch[i][1] ** 2 / batch[i][2], rtol=1e-12)
    # determinism: bitwise identical on repeat
    np.testing.assert_array_equal(c.reweight_eval(betas), batch)
    # ESS decreases with beta past the last stored beta, ESS(w uniform)=N: sanity of the domain properties
    if n > 100:
        w = c.weights(0.37, *batch[3][:2])
        s = c.sum_sq_max(w)
        np.testing.assert_allclose(s[0], 1.0, rtol=1e-10)
        np.testing.assert_allclose(1.0 / s[1], batch[3][1] ** 2 / batch[3][2], rtol=1e-10)


# --------------------------------------------------------------------------- resampling
@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 100_003, 3_000_001])
def test_cdf(dev, n):
    rs = np.random.RandomState(1)
    w = np.exp(rs.randn(n))
    c = ctx_for(1)
    wt = torch.from_numpy(w).to(dev)
    cdf = c.cdf(wt).cpu().numpy()
    np.testing.assert_allclose(cdf, np.cumsum(w), rtol=1e-12)
    assert np.all(np.diff(cdf) >= 0)
    thr = torch.tensor([1.0], dtype=torch.float64, device=dev)
    cdfm = c.cdf(wt, thr).cpu().numpy()
    np.testing.assert_allclose(cdfm, np.cumsum(np.where(w >= 1.0, w, 0.0)), rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("off_in,off_out", [(1, 0), (0, 1), (1, 1), (3, 2)])
def test_cdf_unaligned_views(dev, off_in, off_out):
    """Weights and cdf at addresses that are 8- but not 16-byte aligned (views into larger arrays): the scan's two-values-per-lane
    loads and stores fall back to single ones; same values as the aligned call, bit for bit, and nothing written outside."""
    rs = np.random.RandomState(2)
    n = 5 * 2048 + 77
    w = np.exp(rs.randn(n))
    c = ctx_for(1)
    aligned = c.cdf(torch.from_numpy(w).to(dev)).cpu().numpy()
    buf = torch.zeros(n + 8, dtype=torch.float64, device=dev)
    buf[off_in:off_in + n] = torch.from_numpy(w).to(dev)
    out = torch.full((n + 8,), -7.0, dtype=torch.float64, device=dev)
    c.cdf(buf[off_in:off_in + n], out=out[off_out:off_out + n])
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[off_out:off_out + n], aligned)
    assert np.all(o[:off_out] == -7.0) and np.all(o[off_out + n:] == -7.0)


def test_cdf_wide_dynamic_range(dev):
    """Importance weights span tens of orders of magnitude: the scan must keep the small prefix in front of a
    dominant weight (no `inclusive - own` cancellation) and stay monotone."""
    rs = np.random.RandomState(5)
    n = 300_000
    w = np.exp(rs.randn(n) * 25.0)
    w /= w.sum()
    c = ctx_for(1)
    cdf = c.cdf(torch.from_numpy(w).to(dev)).cpu().numpy()
    ref = np.cumsum(w)
    np.testing.assert_allclose(cdf, ref, rtol=1e-12, atol=0)
    assert np.all(np.diff(cdf) >= -1e-15 * cdf[1:])


def test_systematic_golden(dev):
    g = load("g4_resample.npz")
    c = ctx_for(1)
    for k in range(int(g["n_cases"])):
        w = g[f"c{k}_w"]
        size = int(g[f"c{k}_size"])
        cdf = c.cdf(torch.from_numpy(w).to(dev))
        tot = float(w.sum())
        renorm = tot if abs(tot - 1.0) > ps.SQRTEPS else 1.0
        idx = c.resample_systematic(cdf, size, float(g[f"c{k}_u0"]), renorm=renorm).cpu().numpy()
        np.testing.assert_array_equal(idx, g[f"c{k}_idx"])


def test_multinomial_vs_oracle_and_gather(dev):
    from tempest_amd.device import TAG_RESAMPLE
    rs = np.random.RandomState(3)
    d, nh, n_out = 3, 50_000, 20_000
    u = rs.rand(nh, d); x = 20 * u - 10; logl = -np.sum(x ** 2, axis=1)
    w = np.exp(rs.randn(nh) * 2); w /= w.sum()
    c = ctx_for(d)
    c.history_load(u, x, logl, [0.0], [0.0], [nh])
    cdf = c.cdf(torch.from_numpy(w).to(dev))
    idx = c.resample_multinomial(cdf, n_out, seed=99, tick=5, item0=1000)
    U = px.uniform1(99, np.arange(n_out) + 1000, 5, TAG_RESAMPLE)
    want = ps.multinomial_resample(w, U)
    got = idx.cpu().numpy()
    mism = np.nonzero(got != want)[0]
    assert mism.size <= 2, mism.size     # only draws within 1 ulp of a cdf boundary may differ (scan order)
    uo, xo, lo = c.empty(d, n_out), c.empty(d, n_out), c.empty(n_out)
    c.gather(idx, uo, xo, lo)
    np.testing.assert_array_equal(aos(uo), u[got])
    np.testing.assert_array_equal(aos(xo), x[got])
    np.testing.assert_array_equal(lo.cpu().numpy(), logl[got])
    # distributional: counts track the weights
    counts = np.bincount(got, minlength=nh)
    top = np.argsort(w)[-50:]
    np.testing.assert_allclose(counts[top].sum() / n_out, w[top].sum(), rtol=0.05)


# ------------------------------------------------------------------------------ trimming
def test_trim_threshold_golden(dev):
    g = load("g2_tools.npz")
    c = ctx_for(1)
    for k in range(4):
        w = g[f"ess{k}_w"]
        wn = w / w.sum()
        for j in range(3):
            ess, bins = g[f"trim{k}_{j}_cfg"]
            _, out = c.trim_threshold(torch.from_numpy(wn).to(dev), float(ess), int(bins), sync=True)
            thr, ksum, kcnt, ess_total = out
            mask = wn >= thr
            np.testing.assert_array_equal(np.nonzero(mask)[0], g[f"trim{k}_{j}_idx"])
            assert int(kcnt) == g[f"trim{k}_{j}_idx"].size
            np.testing.assert_allclose(wn[mask] / ksum, g[f"trim{k}_{j}_w"], rtol=1e-12)
            np.testing.assert_allclose(ess_total, ps.effective_sample_size(w), rtol=1e-12)


def test_trim_many_candidates_and_tiny_inputs(dev):
    """The segmented form of the kept sums at its edges: more candidates than rows, a single row, all weights equal (every
    candidate has the same first kept row), 5000 candidates (dynamic LDS above 64 KB) -- against the sorted-array oracle."""
    rs = np.random.RandomState(12)
    c = ctx_for(1)
    cases = [(np.array([1.0]), 0.99, 10), (np.full(7, 1.0 / 7), 0.5, 1000), (rs.rand(50), 0.9, 1000),
             (np.exp(2 * rs.randn(200_000)), 0.99, 5000), (np.exp(rs.randn(40_000)), 0.999, 1)]
    for w, ess, bins in cases:
        w = w / w.sum()
        _, out = c.trim_threshold(torch.from_numpy(w).to(dev), ess, bins, sync=True)
        thr, ksum, kcnt = ps.trim_threshold_sorted(w, ess, bins, normalized=True)
        assert out[0] == thr and int(out[2]) == kcnt, (len(w), bins)
        np.testing.assert_allclose(out[1], ksum, rtol=1e-12)
        np.testing.assert_allclose(out[3], ps.effective_sample_size(w), rtol=1e-12)


def test_trim_large_vs_oracle(dev):
    rs = np.random.RandomState(4)
    w = np.exp(3 * rs.randn(1_500_000)); w /= w.sum()
    c = ctx_for(1)
    _, out = c.trim_threshold(torch.from_numpy(w).to(dev), 0.99, 1000, sync=True)
    thr, ksum, kcnt = ps.trim_threshold_sorted(w, 0.99, 1000, normalized=True)
    assert out[0] == thr and int(out[2]) == kcnt
    np.testing.assert_allclose(out[1], ksum, rtol=1e-12)


# ------------------------------------------------------------------------- proposal fit
def test_fit_modes_vs_golden_and_oracle(dev):
    g = load("g7_modes.npz")
    u, w, idx = g["fg_u"], g["fg_w"], g["fg_idx"]
    n, d = u.shape
    c = ctx_for(d)
    c.history_load(u, u, np.zeros(n), [0.0], [0.0], [n])
    counts = np.bincount(idx, minlength=n).astype(np.int32)
    means, covs, chol, inv, _ = c.fit_modes(torch.from_numpy(counts).to(dev))
    np.testing.assert_allclose(means.cpu().numpy(), g["fg_means"], rtol=1e-14)
    np.testing.assert_allclose(covs.cpu().numpy(), g["fg_covs"], rtol=1e-10)
    np.testing.assert_allclose(chol.cpu().numpy(), g["fg_chol"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(inv.cpu().numpy(), g["fg_inv"], rtol=1e-8)
    # per-label fit (modes.py:131-219)
    labels = g["fp_labels"].astype(np.int32)
    counts = np.zeros(n, dtype=np.int32)
    for cidx in range(3):
        sel = np.nonzero(labels == cidx)[0]
        counts += np.bincount(sel[g[f"fp_idx{cidx}"]], minlength=n).astype(np.int32)
    means, covs, chol, inv, _ = c.fit_modes(torch.from_numpy(counts).to(dev), torch.from_numpy(labels).to(dev), K=3)
    np.testing.assert_allclose(means.cpu().numpy(), g["fp_means"], rtol=1e-14)
    np.testing.assert_allclose(covs.cpu().numpy(), g["fp_covs"], rtol=1e-10)
    np.testing.assert_allclose(inv.cpu().numpy(), g["fp_inv"], rtol=1e-8)


def test_fit_modes_large_narrow_with_duplicates(dev):
    rs = np.random.RandomState(8)
    n, d = 300_000, 10
    base = 0.5 + 2e-3 * rs.randn(n // 3, d) @ np.triu(rs.rand(d, d))
    u = np.concatenate([base, base, base])[rs.permutation(n)]     # exact duplicates, as rejected moves leave
    u[:, 0] = np.clip(np.abs(1e-9 * rs.randn(n)), 0, 1)            # a coordinate piled against the bound
    w = np.exp(rs.randn(n)); w /= w.sum()
    c = ctx_for(d)
    c.history_load(u, u, np.zeros(n), [0.0], [0.0], [n])
    cdf = c.cdf(torch.from_numpy(w).to(dev))
    counts = c.multinomial_counts(cdf, seed=5, tick=1, factor=4)
    cn = counts.cpu().numpy()
    assert cn.sum() == 4 * n
    U = px.uniform1(5, np.arange(4 * n), 1, px.TAG_UPSAMPLE)
    want_counts = np.bincount(ps.multinomial_resample(w, U), minlength=n)
    assert np.abs(cn - want_counts).sum() <= 4
    means, covs, chol, inv, _ = c.fit_modes(counts)
    mu, Sig, _ = ps.median_cov_from_counts(u, cn)
    np.testing.assert_allclose(means.cpu().numpy()[0], mu, rtol=1e-14, atol=0)
    np.testing.assert_allclose(covs.cpu().numpy()[0], Sig, rtol=1e-9, atol=1e-22)
    L = chol.cpu().numpy()[0]
    np.testing.assert_allclose(L @ L.T, covs.cpu().numpy()[0], rtol=1e-10, atol=1e-22)


@pytest.mark.parametrize("kept", [True, False])
def test_multinomial_counts_sorted_draws_equal_the_draw_order_lookups(dev, kept):
    """TPH_OPT_SORTED_DRAWS: >= 2^21 draws are sorted and counted by the owners of the cdf's tiles (k_mc_tiles) instead of looked
    up one by one -- the same counts, row by row, on a heavy-tailed trimmed weight vector (flat stretches of the cdf where rows are trimmed away, a
    few rows that take thousands of draws), with the number of draws on the device (kept rows x 4) or given."""
    from tempest_amd.device import OPT_SORTED_DRAWS
    rs = np.random.RandomState(77)
    n = 6_000_000
    w = np.exp(2.0 * rs.randn(n))
    w[rs.randint(n, size=20)] *= 1e3                   # a few dominant rows
    w[rs.rand(n) < 0.3] = 0.0                           # rows without weight
    w /= w.sum()
    c = ctx_for(4)
    wt = torch.from_numpy(w).to(dev)
    thr = c.trim_threshold(wt, ess=0.99)
    cdf = c.cdf(wt, thr=thr[0:1])
    kc = thr[2:3] if kept else None
    nd = 4 * n if kept else 9_000_000
    got = {}
    for mode in (1, 0):
        c.set_option(OPT_SORTED_DRAWS, mode)
        got[mode] = c.multinomial_counts(cdf, seed=11, tick=3, kept_count=kc, factor=4, n_draw_max=nd).cpu().numpy()
    c.set_option(OPT_SORTED_DRAWS, 1)
    n_draw = 4 * int(thr[2].item()) if kept else nd
    assert n_draw >= 1 << 21, n_draw                    # the sorted path really ran
    assert got[1].sum() == n_draw
    np.testing.assert_array_equal(got[1], got[0])
    assert got[1].max() > 1000 and (got[1] == 0).mean() > 0.3


def test_multinomial_counts_sorted_draws_edge_cases(dev):
    """The count over sorted draws at its edges (threshold lowered to 2 draws through the option): a handful of draws, a count
    that is no multiple of anything, many more draws than buckets of the sort (several per bucket: the margins of a tile's stretch
    matter); all the mass on one row, on the last row of a tile, on the first row of the next, on the very last row; weightless
    rows at both ends of the cdf; row counts of exactly one tile, one more, two tiles, two and one; two rows; a single row.
    Always the counts of the draw-order lookups."""
    from tempest_amd.device import OPT_SORTED_DRAWS
    rs = np.random.RandomState(5)
    c = ctx_for(3)
    cases = []
    w = rs.rand(1000); cases.append(w / w.sum())
    w = np.zeros(5000); w[1234] = 1.0; cases.append(w)
    w = rs.rand(4096); w[:700] = 0.0; w[-900:] = 0.0; cases.append(w / w.sum())
    cases.append(np.array([0.25, 0.75]))
    cases.append(np.array([1.0]))
    w = np.exp(8.0 * rs.randn(20000)); cases.append(w / w.sum())
    for n in (2048, 2049, 4096, 4097):
        w = rs.rand(n) ** 4; cases.append(w / w.sum())
    for hot in (2047, 2048, 6143):
        w = 1e-7 * rs.rand(6144); w[hot] = 1.0; cases.append(w / w.sum())
    w = np.zeros(6144); w[2047] = 0.5; w[2048] = 0.5; cases.append(w)
    cases.append(np.zeros(5000))                        # no weight at all (a degenerate cdf): every draw counts for the last row
    for w in cases:
        cdf = c.cdf(torch.from_numpy(w).to(dev))
        for n_draw in (2, 15, 16, 17, 1000, 4099, 300_001):
            got = {}
            for mode in (2, 0):
                c.set_option(OPT_SORTED_DRAWS, mode)
                got[mode] = c.multinomial_counts(cdf, seed=3, tick=n_draw, kept_count=None, factor=1, n_draw_max=n_draw).cpu().numpy()
            assert got[2].sum() == n_draw
            np.testing.assert_array_equal(got[2], got[0])
            if w.sum() > 0:
                assert np.all(got[2][w == 0.0] == 0)
            else:
                assert got[2][-1] == n_draw
    kc = torch.tensor([123.0], dtype=torch.float64, device=dev)          # the draw count from the device: 4 x 123
    cdf = c.cdf(torch.from_numpy(cases[0]).to(dev))
    c.set_option(OPT_SORTED_DRAWS, 2)
    a = c.multinomial_counts(cdf, seed=3, tick=9, kept_count=kc, factor=4, n_draw_max=4000).cpu().numpy()
    c.set_option(OPT_SORTED_DRAWS, 0)
    b = c.multinomial_counts(cdf, seed=3, tick=9, kept_count=kc, factor=4, n_draw_max=4000).cpu().numpy()
    c.set_option(OPT_SORTED_DRAWS, 1)
    assert a.sum() == 492
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("d,K", [(10, 1), (10, 3), (37, 1)])
def test_fit_modes_compacts_sparse_upsampled_sets(dev, d, K):
    """Histories >= 262 144 rows whose multiplicities are mostly zero go through the order-preserving stream compaction
    (k_nz_count / k_nz_offsets / k_nz_scatter) before the five fit passes: same medians (exact) and covariances."""
    rs = np.random.RandomState(100 + d + K)
    n = 300_001
    u = np.clip(0.5 + 0.05 * rs.randn(n, d) @ np.triu(rs.rand(d, d)) / np.sqrt(d), 0, 1)
    counts = np.zeros(n, dtype=np.int32)
    hot = rs.choice(n, n // 9, replace=False)                     # 11 % of the rows carry all the multiplicity
    counts[hot] = rs.randint(1, 9, size=hot.size)
    counts[-1] = 3; counts[0] = 2                                  # first and last row of the history kept
    labels = rs.randint(0, K, size=n).astype(np.int32)
    c = ctx_for(d)
    c.history_load(u, u, np.zeros(n), [0.0], [0.0], [n])
    ct = torch.from_numpy(counts).to(dev)
    lt = torch.from_numpy(labels).to(dev) if K > 1 else None
    means, covs, chol, inv, _ = c.fit_modes(ct, lt, K=K)
    for k in range(K):
        ck = counts * (labels == k) if K > 1 else counts
        mu, Sig, _ = ps.median_cov_from_counts(u, ck)
        np.testing.assert_allclose(means.cpu().numpy()[k], mu, rtol=1e-14, atol=0)
        np.testing.assert_allclose(covs.cpu().numpy()[k], Sig, rtol=1e-9, atol=1e-22)
    # dense multiplicities (> half of the rows): the history is streamed as it is -- same answer as the oracle too
    dense = rs.randint(0, 3, size=n).astype(np.int32)
    means, covs, _, _, _ = c.fit_modes(torch.from_numpy(dense).to(dev))
    mu, Sig, _ = ps.median_cov_from_counts(u, dense)
    np.testing.assert_allclose(means.cpu().numpy()[0], mu, rtol=1e-14, atol=0)
    np.testing.assert_allclose(covs.cpu().numpy()[0], Sig, rtol=1e-9, atol=1e-22)


def test_chol_inv_ridge(dev):
    rs = np.random.RandomState(2)
    d = 6
    A = rs.randn(d, d); good = A @ A.T + 0.1 * np.eye(d)
    v = rs.randn(d, 1); singular = v @ v.T                       # rank 1 -> LinAlgError -> ridge
    covs = np.stack([good, singular, np.zeros((d, d))])
    want_cov, want_chol, want_inv = ps.mode_statistics(np.zeros((3, d)), covs)
    c = ctx_for(d)
    ct = torch.from_numpy(covs.copy()).to(dev)
    chol, inv, winv = c.chol_inv(ct)
    np.testing.assert_allclose(ct.cpu().numpy(), want_cov, rtol=1e-14)
    np.testing.assert_allclose(chol.cpu().numpy(), want_chol, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(inv.cpu().numpy()[0], want_inv[0], rtol=1e-9)
    np.testing.assert_allclose(inv.cpu().numpy()[2], want_inv[2], rtol=1e-9)
    # L^-1 (what the proposal kernels consume): lower-triangular, W L = I, W^T W = Sigma^-1
    W, L = winv.cpu().numpy(), chol.cpu().numpy()
    for k in (0, 2):
        assert np.allclose(np.triu(W[k], 1), 0.0)
        np.testing.assert_allclose(W[k] @ L[k], np.eye(d), atol=1e-9)
        np.testing.assert_allclose(W[k].T @ W[k], want_inv[k], rtol=1e-8, atol=1e-12)


def test_volume_variation_golden(dev):
    g = load("g2_tools.npz")
    for key in ("vv0", "vv4"):
        x, w = g[f"{key}_x"], g[f"{key}_w"]
        n, d = x.shape
        wn = w / w.sum()
        c = ctx_for(d)
        c.history_load(x, x, np.zeros(n), [0.0], [0.0], [n])
        wt = torch.from_numpy(wn).to(dev)
        mc = c.weighted_moments(wt).cpu().numpy()
        mean, cov = mc[:d], mc[d:].reshape(d, d)
        np.testing.assert_allclose(mean, np.sum(x * wn[:, None], axis=0), rtol=1e-12)
        xc = x - mean
        np.testing.assert_allclose(cov, xc.T @ (xc * wn[:, None]), rtol=1e-9, atol=1e-20)
        assert np.linalg.matrix_rank(cov) == d
        cinv = np.linalg.inv(cov)
        s = c.cv_sum(wt, torch.from_numpy(mean).to(dev), torch.from_numpy(cinv).to(dev)).cpu().numpy()[0]
        np.testing.assert_allclose(0.5 * np.sqrt(s), g[key], rtol=1e-8)


# ----------------------------------------------------------------------------- mutation
class _Modes:
    def __init__(self, means, chol, inv, dof, dev):
        self.K = means.shape[0]
        self.means_dev = torch.from_numpy(means).to(dev)
        self.chol_dev = torch.from_numpy(chol).to(dev)
        self.inv_dev = torch.from_numpy(inv).to(dev)
        self.dof_dev = torch.from_numpy(dof).to(dev)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("variant", ["multilane_d7", "registers_d7", "generic_d7", "generic_d19", "multilane_d19",
                                     "multilane_d50", "multilane_d3"])
def test_propose_accept_adapt_vs_oracle(dev, kernel, bc, variant):
    rs = np.random.RandomState(17)
    d, n, K = int(variant.split("_d")[1]), 5000, 3
    means = 0.5 + 0.1 * rs.randn(K, d)
    covs = np.empty((K, d, d))
    for k in range(K):
        A = rs.randn(d, d) * 0.08
        covs[k] = A @ A.T + 1e-3 * np.eye(d)
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([1e6, 4.0, 25.0])
    sigmas = np.array([0.9, 0.5, 0.2]) * (2.38 / np.sqrt(d) if kernel == "rwm" else 1.0)
    assign = rs.randint(K, size=n).astype(np.int32)
    u = np.clip(means[assign] + 0.2 * rs.randn(n, d), 0.01, 0.99)   # many land near the walls -> redraws
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 4242, 11, 100_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    c = ctx_for(d)
    # TPH_OPT_PROPOSE_VARIANT: 1 = one-lane LDS kernel, 2 = one-lane register kernel (d<=16), 3 = multi-lane kernel
    c.set_option(0, {"multilane": 3, "generic": 1, "registers": 2}[variant.split("_")[0]])
    modes = _Modes(means, chol, inv, dof, dev)
    up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
    ut = soa(u, dev)
    at = torch.from_numpy(assign).to(dev)
    st = torch.from_numpy(sigmas).to(dev)
    ft = torch.from_numpy(flags).to(dev)
    c.propose(kernel, ut, at, modes, st, ft, seed, tick, item0, up, mu_, mup)
    got_up = aos(up)
    np.testing.assert_allclose(got_up, want_up, rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((got_up[:, strict] >= 0) & (got_up[:, strict] <= 1))
    assert np.all((got_up >= 0) & (got_up <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(mu_.cpu().numpy(), want_mu, rtol=1e-10)
        np.testing.assert_allclose(mup.cpu().numpy(), want_mup, rtol=1e-9, atol=1e-9)
    # accept
    x = 20 * u - 10
    logl = -0.5 * np.sum(x ** 2, axis=1) * 0.05
    xp = 20 * want_up - 10
    loglp = -0.5 * np.sum(xp ** 2, axis=1) * 0.05
    loglp[2::103] = np.inf; loglp[1::101] = -np.inf; loglp[::97] = np.nan
    beta = 0.6
    alpha, mask = omc.accept(kernel, beta, logl, loglp, want_mu, want_mup, dof, assign, d, seed, tick + 1, item0)
    xt, lt = soa(x, dev), torch.from_numpy(logl.copy()).to(dev)
    upt, xpt, lpt = soa(want_up, dev), soa(xp, dev), torch.from_numpy(loglp).to(dev)
    sums = c.empty(1 + K)
    c.accept(kernel, beta, ut, xt, lt, upt, xpt, lpt, torch.from_numpy(want_mu).to(dev),
             torch.from_numpy(want_mup).to(dev), at, K, modes.dof_dev, seed, tick + 1, item0, sums)
    s = sums.cpu().numpy()
    assert s[0] == mask.sum()
    for k in range(K):
        np.testing.assert_allclose(s[1 + k], alpha[assign == k].sum(), rtol=1e-10)
    np.testing.assert_array_equal(aos(ut), np.where(mask[:, None], want_up, u))
    np.testing.assert_array_equal(aos(xt), np.where(mask[:, None], xp, x))
    np.testing.assert_array_equal(lt.cpu().numpy(), np.where(mask, loglp, logl))
    assert not mask[::97].any() and not mask[1::101].any()
    # adapt + stopping rule
    counts = c.cluster_counts(at, n, K)
    np.testing.assert_array_equal(counts.cpu().numpy(), np.bincount(assign, minlength=K))
    state = c.zeros(6)
    state[0] = 4.0
    c.adapt(kernel, sums, counts, K, n, 2, 40, st, state)
    ws, done, acc, target = omc.adapt(kernel, alpha, mask, assign, K, sigmas, 5, d, 2, 40)
    np.testing.assert_allclose(st.cpu().numpy(), ws, rtol=1e-12)
    sh = state.cpu().numpy()
    assert sh[0] == 5 and bool(sh[1]) == bool(done) and int(sh[5]) == target
    np.testing.assert_allclose(sh[2], acc, rtol=1e-14)
    np.testing.assert_allclose(sh[3], alpha.mean(), rtol=1e-10)
    np.testing.assert_allclose(sh[4], ws.mean() / (2.38 / np.sqrt(d)), rtol=1e-12)


def test_propose_golden_pure_function(dev):
    # single-cluster-per-lane golden vectors drawn by the reference itself: check the deterministic
    # parts (Mahalanobis, acceptance factor) against its outputs
    g = load("g6_mcmc.npz")
    u, a = g["u"], g["assign"].astype(np.int32)
    n, d = u.shape
    c = ctx_for(d)
    modes = _Modes(g["means"], g["chol"], g["inv"], g["dof"], dev)
    up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
    c.propose("tpcn", soa(u, dev), torch.from_numpy(a).to(dev), modes, torch.from_numpy(g["tpcn_sigmas"]).to(dev),
              None, 1, 1, 0, up, mu_, mup)
    np.testing.assert_allclose(mu_.cpu().numpy(), ps.mahalanobis(u, g["means"], g["inv"], a), rtol=1e-10)
    fac = ps.tpcn_acceptance_factor(u, aos(up), g["means"], g["inv"], g["dof"], a)
    nu = g["dof"][a]
    dev_fac = 0.5 * (d + nu) * (np.log1p(mup.cpu().numpy() / nu) - np.log1p(mu_.cpu().numpy() / nu))
    np.testing.assert_allclose(dev_fac, fac, rtol=1e-6, atol=1e-9)


def test_inf_repair_vs_oracle(dev):
    rs = np.random.RandomState(6)
    d, n = 3, 10_000
    u = rs.rand(n, d); x = 20 * u - 10
    logl = -0.5 * np.sum(x ** 2, axis=1)
    logl[x[:, 0] > 5] = -np.inf
    logl[x[:, 1] < -8] = np.inf
    wu, wx, wl, nfin = omc.inf_repair(u, x, logl, 77, 3, 500)
    c = ctx_for(d)
    ut, xt, lt = soa(u, dev), soa(x, dev), torch.from_numpy(logl.copy()).to(dev)
    stats = c.inf_repair(ut, xt, lt, 77, 3, 500).cpu().numpy()
    assert stats[0] == nfin and stats[1] == n
    np.testing.assert_array_equal(aos(ut), wu)
    np.testing.assert_array_equal(aos(xt), wx)
    np.testing.assert_array_equal(lt.cpu().numpy(), wl)
    assert np.all(np.isfinite(lt.cpu().numpy()))


@pytest.mark.parametrize("d,n,m", [(3, 1000, 1), (10, 5000, 777), (37, 3000, 3000), (100, 2000, 129)])
def test_posterior_rows_and_index_compose(d, n, m):
    """tph_posterior_rows: selected history rows as row-major (m, d) + logl + w / wdiv; tph_index_compose: a[b]."""
    from tempest_amd.device import HipContext, KEY_U, KEY_X
    rng = np.random.RandomState(d)
    ctx = HipContext(d, device=0)
    u, x, logl = rng.rand(n, d), rng.randn(n, d), rng.randn(n)
    ctx.history_load(u, x, logl, [0.0], [0.0], [n])
    w = torch.from_numpy(rng.rand(n)).cuda()
    idx_h = np.sort(rng.choice(n, m, replace=False)) if m < n else rng.permutation(n)
    idx = torch.from_numpy(idx_h.astype(np.int64)).cuda()
    xo, lo, wo = ctx.posterior_rows(idx, m, w=w, wdiv=0.37)
    np.testing.assert_array_equal(xo.cpu().numpy(), x[idx_h])
    np.testing.assert_array_equal(lo.cpu().numpy(), logl[idx_h])
    np.testing.assert_array_equal(wo.cpu().numpy(), w.cpu().numpy()[idx_h] / 0.37)
    uo, lo2, none = ctx.posterior_rows(None, m, key=KEY_U)
    assert none is None
    np.testing.assert_array_equal(uo.cpu().numpy(), u[:m])
    np.testing.assert_array_equal(lo2.cpu().numpy(), logl[:m])
    b_h = rng.randint(0, m, size=2 * m + 3)
    comp = ctx.index_compose(idx, torch.from_numpy(b_h.astype(np.int64)).cuda())
    np.testing.assert_array_equal(comp.cpu().numpy(), idx_h[b_h])
    _ = KEY_X


@pytest.mark.parametrize("d", [1, 5, 12])
def test_weighted_moments_shifted_one_pass(dev, d):
    """tph_weighted_moments_shifted: weighted mean and covariance in one pass about a nearby centre == NumPy's two-pass values,
    also for a narrow cloud far from the origin of the unit cube (where raw second moments would cancel)."""
    rs = np.random.RandomState(d)
    n = 200_003
    u = np.clip(0.83 + 1e-4 * rs.randn(n, d), 0, 1)                  # |mean|^2 / var ~ 7e7
    w = np.exp(rs.randn(n)); w /= w.sum()
    c = ctx_for(d)
    c.history_load(u, u, np.zeros(n), [0.0], [0.0], [n])
    wt = torch.from_numpy(w).to(dev)
    centre = torch.from_numpy(u[0].copy()).to(dev)
    out = c.weighted_moments_shifted(wt, centre).cpu().numpy()
    mean = (u * w[:, None]).sum(axis=0)
    xc = u - mean
    cov = xc.T @ (xc * w[:, None])
    np.testing.assert_allclose(out[0], w.sum(), rtol=1e-13)
    np.testing.assert_allclose(out[1:1 + d], mean, rtol=1e-13)
    np.testing.assert_allclose(out[1 + d:].reshape(d, d), cov, rtol=1e-9, atol=1e-22)


# ------------------------------------------------------------------------------------------------------------------
# global entry points of the sharded path through a ONE-RANK loopback communicator: all collectives become identities, the
# blocked / staged kernels run, and the results must equal the plain one-GPU functions (up to summation order).
def _history_ctx(d, T, rows, seed, loopback):
    from tempest_amd.comm import attach_loopback
    from tempest_amd.device import HipContext
    rs = np.random.RandomState(seed)
    n = T * rows
    c = HipContext(d, 0)
    u = rs.rand(n, d)
    logl = -rs.chisquare(5, size=n) * 3.0
    c.history_load(u, 20 * u - 10, logl, np.linspace(0, 0.5, T), -np.arange(T, dtype=float), [rows] * T)
    if loopback:
        attach_loopback(c, p2p=(loopback == "p2p"))
    return c, rs


@pytest.mark.parametrize("T,rows,p2p", [(1, 5000, False), (7, 3000, True), (40, 700, True), (3, 70000, False), (17, 65536, True)])
def test_global_entry_points_loopback_match_plain(dev, T, rows, p2p):
    d = 3
    plain, rs = _history_ctx(d, T, rows, 11, False)
    loop, _ = _history_ctx(d, T, rows, 11, "p2p" if p2p else True)
    assert loop.p2p_active == p2p
    n = T * rows
    w = np.exp(rs.randn(n) * 3.0)
    w /= w.sum()
    wt = torch.from_numpy(w).to(dev)
    # reweight triples through the gathered merge
    a, b = plain.reweight_eval([0.0, 0.4, 1.0]), loop.reweight_eval([0.0, 0.4, 1.0])
    np.testing.assert_allclose(a[:, 0] + np.log(a[:, 1]), b[:, 0] + np.log(b[:, 1]), rtol=1e-13)
    np.testing.assert_allclose(a[:, 1] ** 2 / a[:, 2], b[:, 1] ** 2 / b[:, 2], rtol=1e-12)
    # trim threshold: the radix descent finds the same order statistics -> the same threshold, bit for bit
    for ess, bins in ((0.99, 1000), (0.9, 37), (0.5, 1)):
        tp_ = plain.trim_threshold(wt, ess, bins).cpu().numpy()
        tl_ = loop.trim_threshold(wt, ess, bins, global_=True).cpu().numpy()
        assert tp_[0] == tl_[0] and tp_[2] == tl_[2], (ess, bins, tp_, tl_)
        np.testing.assert_allclose(tl_[[1, 3]], tp_[[1, 3]], rtol=1e-12)
    thr = plain.trim_threshold(wt, 0.99, 1000)
    # cumulative weights (plain and masked), totals
    for t in (None, thr[0:1]):
        cp = plain.cdf(wt, t).cpu().numpy()
        cl, tot = loop.cdf_global(wt, t, total=True)
        cl = cl.cpu().numpy()
        np.testing.assert_allclose(cl, cp, rtol=1e-12, atol=1e-300)
        assert tot == cl[-1] and np.all(np.diff(cl) >= -4e-16 * tot)      # monotone up to rounding (scan.h)
        assert np.all(cl[rows - 1::rows][:-1] <= cl[rows::rows] + 0)       # block ends meet the next block's start
    # draws: the same rows (a draw within one rounding error of a boundary may move to the neighbouring row)
    cp, cl = plain.cdf(wt), loop.cdf_global(wt)
    ip = plain.resample_multinomial(cp, 20000, 99, 3).cpu().numpy()
    il = loop.resample_select_global(cl, 20000, 0, 99, 3).cpu().numpy()
    assert (ip != il).sum() <= 2 and np.abs(ip - il).max() <= 1
    tot = float(cp[-1].item())
    ip = plain.resample_systematic(cp, 7777, 0.37, renorm=tot).cpu().numpy()
    il = loop.resample_select_global(cl, 7777, 1, 99, 3, u0=0.37, pscale=tot).cpu().numpy()
    assert (ip != il).sum() <= 2 and np.abs(ip - il).max() <= 1
    cpm, clm = plain.cdf(wt, thr[0:1]), loop.cdf_global(wt, thr[0:1])
    kp = plain.multinomial_counts(cpm, 5, 8, kept_count=thr[2:3], factor=4, n_draw_max=4 * n).cpu().numpy()
    kl = loop.multinomial_counts_global(clm, 5, 8, kept_count=thr[2:3], factor=4, n_draw_max=4 * n).cpu().numpy()
    assert kp.sum() == kl.sum() == 4 * int(thr[2].item()) and np.abs(kp - kl).sum() <= 4
    # the fit on identical multiplicities: medians exactly, moments to rounding
    counts = torch.from_numpy(kp).to(dev)
    fp = plain.fit_modes(counts)
    fl = loop.fit_modes(counts, global_=True)
    assert torch.equal(fp[0], fl[0])
    for x, y, tol in zip(fp[1:], fl[1:], (1e-11, 1e-9, 1e-8, 1e-8)):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=tol, atol=1e-14)
    labels = torch.from_numpy((rs.rand(n) < 0.4).astype(np.int32)).to(dev)
    fp = plain.fit_modes(counts, labels, K=2)
    fl = loop.fit_modes(counts, labels, K=2, global_=True)
    assert torch.equal(fp[0], fl[0])
    np.testing.assert_allclose(fl[1].cpu().numpy(), fp[1].cpu().numpy(), rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("variant", ["registers_d7", "multilane_d7", "generic_d7", "multilane_d19", "registers_d10"])
def test_deferred_metropolis_update_equals_in_place(dev, kernel, variant):
    """tph_accept in deferred mode (decision -> pending mask, u untouched) followed by tph_propose with that mask (accepted
    proposals moved into place, then proposed from) is the same chain, bit for bit, as the in-place update: same proposals,
    same log-likelihoods and Mahalanobis forms after every step, and the same u once the last mask has been resolved."""
    rs = np.random.RandomState(23)
    d, n, K = int(variant.split("_d")[1]), 70_000 if variant == "registers_d10" else 6000, 2
    means = 0.5 + 0.05 * rs.randn(K, d)
    covs = np.empty((K, d, d))
    for k in range(K):
        A = rs.randn(d, d) * 0.07
        covs[k] = A @ A.T + 1e-3 * np.eye(d)
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([1e6, 6.0])
    sigmas = np.array([0.7, 0.4]) * (2.38 / np.sqrt(d) if kernel == "rwm" else 1.0)
    assign = rs.randint(K, size=n).astype(np.int32)
    u0 = np.clip(means[assign] + 0.15 * rs.randn(n, d), 0.01, 0.99)
    c = ctx_for(d)
    c.set_option(0, {"multilane": 3, "generic": 1, "registers": 2}[variant.split("_")[0]])
    modes = _Modes(means, chol, inv, dof, dev)
    at, st = torch.from_numpy(assign).to(dev), torch.from_numpy(sigmas).to(dev)

    def like(up):
        x = 20 * up - 10
        return -0.5 * (x * x).sum(dim=0) * 0.05

    def chain(deferred):
        u = soa(u0, dev)
        logl = like(u).clone()
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        sums = c.empty(1 + K)
        pend = torch.zeros(n, dtype=torch.uint8, device=dev) if deferred else None
        trace = []
        for step in range(4):
            tick = 50 + 2 * step
            c.propose(kernel, u, at, modes, st, None, 99, tick, 7, up, mu_, mup, pending=pend)
            lp = like(up)
            c.accept(kernel, 0.8, u, None, logl, up, None, lp, mu_, mup, at, K, modes.dof_dev, 99, tick + 1, 7, sums, pending=pend)
            trace.append((up.clone(), logl.clone(), mu_.clone(), sums.clone()))
        if deferred:
            assert int(pend.sum().item()) > 0                 # moves are waiting
            c.propose(kernel, u, at, modes, st, None, 99, 999, 7, up, mu_, mup, pending=pend)   # any later proposal resolves them
            assert int(pend.sum().item()) == 0
        return u, trace
    ua, ta = chain(False)
    ub, tb = chain(True)
    for a, b in zip(ta, tb):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(ua, ub)
    assert 0.05 < float(ta[-1][3][0]) / n < 0.95            # a real mix of accepted and rejected moves


@pytest.mark.parametrize("d", [3, 10, 19, 50])
def test_volume_variation_one_call_vs_oracle(dev, d):
    """tph_volume_variation (moments, rank rule, Cholesky / inverse and the blocked triangular sum on the device) against the
    oracle's tools.py:58-117 restatement: full-rank ensembles, a rank-deficient one (ridge branch), n < d + 1."""
    from tempest_amd import tools
    rs = np.random.RandomState(100 + d)
    n = 20000
    A = rs.randn(d, d) * 0.1
    x = 0.5 + rs.randn(n, d) @ A.T
    w = np.exp(rs.randn(n) * 1.5)
    got = tools.volume_variation(x, w)
    np.testing.assert_allclose(got, ps.volume_variation(x, w), rtol=1e-8)
    # a second call on the same context re-uses the previous mean as the centre of the one-pass moments (d <= 12)
    w2 = np.exp(rs.randn(n) * 0.5)
    np.testing.assert_allclose(tools.volume_variation(x, w2), ps.volume_variation(x, w2), rtol=1e-8)
    # rank-deficient: a coordinate repeated exactly -> cov += 1e-6 trace I (tools.py:102-104)
    xd = x.copy()
    xd[:, -1] = xd[:, 0]
    wn = w / w.sum()
    xc = xd - np.sum(xd * wn[:, None], axis=0)
    assert np.linalg.matrix_rank(xc.T @ (xc * wn[:, None])) < d
    np.testing.assert_allclose(tools.volume_variation(xd, w), ps.volume_variation(xd, w), rtol=1e-6)
    # fewer rows than d + 1
    assert tools.volume_variation(x[:d], w[:d]) == 1e10


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("d,rounds", [(19, 0), (33, 0), (50, 0), (65, 0), (100, 0),     # 4 / 8 / 8 / 16 / 16 waves per tile
                                      (33, 2), (50, 24), (65, 3), (100, 6), (112, 2), (17, 3)])   # TPH_OPT_BLOCKED = rounds of the kernel
@pytest.mark.parametrize("mfma,tries", [(1, 1), (1, 3), (0, 1)])
def test_blocked_proposal_kernel_vs_oracle_and_multilane(dev, kernel, bc, d, rounds, mfma, tries):
    """TPH_OPT_PROPOSE_VARIANT 4: attempt 0 of every particle in the blocked kernel -- on the FP64 matrix cores (mfma = 1,
    propose_blkm.hip: a wave per 16 particles, both triangular products as v_mfma_f64_16x16x4 tiles) or with lane = particle
    and L and L^-1 through the scalar cache (mfma = 0, TPH_OPT_BLK_MFMA) --, further rounds of it (attempt 1, 2, ... of the particles still out of bounds, compacted lists), and whoever
    is left finished by the multi-lane kernel.  Same draws and formulas as the other kernels: the proposals and both
    Mahalanobis forms equal the oracle's (and the multi-lane kernel's) to rounding -- on an ensemble where a good share of the
    first attempts fail, so that the later rounds and the straggler pass are exercised."""
    if mfma == 0 and d > 100:
        pytest.skip("the scalar-cache kernel serves n_dim <= 100")
    rs = np.random.RandomState(31 + d)
    n = 3000
    means = 0.5 + 0.05 * rs.randn(1, d)
    A = rs.randn(d, d) * (0.05 / np.sqrt(d))
    covs = (A @ A.T + 2e-4 * np.eye(d))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([8.0])
    sigmas = np.array([0.6]) * (2.38 / np.sqrt(d) if kernel == "rwm" else 1.0)
    assign = np.zeros(n, dtype=np.int32)
    u = np.clip(means[0] + 0.12 * rs.randn(n, d), 0.002, 0.998)
    u[: n // 3, rs.randint(d)] = 0.001                                   # a third of the ensemble sits on a wall
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 777, 5, 4_000_000_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    got = {}
    for variant in (4, 3):
        c = ctx_for(d)
        c.set_option(0, variant)
        c.set_option(4, rounds if variant == 4 else 0)
        c.set_option(15, mfma)                 # TPH_OPT_BLK_MFMA
        c.set_option(16, tries)                # TPH_OPT_BLK_TRIES: attempts per round, in place (matrix-core kernel)
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[variant] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy())
        c.set_option(4, 0)
        c.set_option(15, 1)
    np.testing.assert_allclose(got[4][0], want_up, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(got[4][0], got[3][0], rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((got[4][0][:, strict] >= 0) & (got[4][0][:, strict] <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(got[4][1], want_mu, rtol=1e-9)
        np.testing.assert_allclose(got[4][2], want_mup, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(got[4][2], got[3][2], rtol=1e-8, atol=1e-8)
    # the straggler pass really ran: some first attempts (counter-based draws of attempt 0) are out of bounds
    z0 = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, omc.bc_flags(d, list(range(d)), []), seed, tick, item0)[0]
    failed = np.mean(np.any(np.abs(z0 - want_up) > 1e-9, axis=1))
    assert failed > 0.05
    if rounds > 1 and bc is None and tries == 1:      # with later rounds the redraw probe counts ALL first attempts: n / (n - failures)
        np.testing.assert_allclose(got[4][3][8], 1.0 / (1.0 - failed), rtol=1e-12)
    if rounds > 1 and bc is None and tries > 1:       # ... of the particles whose first `tries` attempts all failed: f = fs^(1/tries)
        assert 1.0 < got[4][3][8] < 2.0 / (1.0 - failed)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("d,rounds,tries,spread,n", [(33, 2, 1, 0.2, 5000), (50, 3, 3, 0.25, 4097), (100, 5, 1, 0.16, 3000), (24, 2, 2, 3.0, 900)])
def test_blocked_list_rounds_fan_out(dev, kernel, d, rounds, tries, spread, n):
    """TPH_OPT_BLK_FAN: a round over a LIST gives every listed particle G consecutive attempts side by side in its tile (G chosen
    on the device from the list's length) and the first in bounds in attempt order wins; the attempt the next round -- and the
    screened straggler pass -- goes on from travels through device memory.  Same attempts win as with one attempt per column and
    round, the proposals equal to rounding and equal to the oracle's; on ensembles where
    most first attempts fail, incl. one (spread 3) where nearly every particle runs into the cap of 256 attempts."""
    rs = np.random.RandomState(5 + d)
    means = np.full((1, d), 0.5)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (spread ** 2) / 2.0)[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([1e6])
    sigmas = np.array([min(2.38 / np.sqrt(d), 0.99)])
    assign = np.zeros(n, dtype=np.int32)
    u = np.clip(0.5 + 0.2 * rs.randn(n, d), 0.001, 0.999)
    flags = omc.bc_flags(d)
    seed, tick, item0 = 4242, 9, 123
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    got = {}
    c = ctx_for(d)
    for fan in (1, 0):
        c.set_option(0, 4); c.set_option(4, rounds); c.set_option(15, 1); c.set_option(16, tries); c.set_option(17, fan)
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[fan] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy())
    c.set_option(0, 0); c.set_option(4, 0); c.set_option(16, 0); c.set_option(17, 1)
    # (to rounding, not bit for bit: WHICH kernel evaluates a particle's winning attempt -- a matrix-core round or the screened
    # kernel's FP64 chain -- depends on the round that settles it, and the two sum a row in different orders)
    np.testing.assert_allclose(got[1][0], got[0][0], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(got[1][2], got[0][2], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(np.any(got[1][0] != u, axis=1), np.any(got[0][0] != u, axis=1))
    np.testing.assert_allclose(got[1][0], want_up, rtol=1e-11, atol=1e-13)
    if kernel == "tpcn":
        np.testing.assert_allclose(got[1][2], want_mup, rtol=1e-8, atol=1e-8)
    moved = np.any(np.abs(got[1][0] - u) > 1e-12, axis=1)
    z0 = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, omc.bc_flags(d, list(range(d)), []), seed, tick, item0)[0]
    failed_first = np.mean(np.any(np.abs(z0 - want_up) > 1e-9, axis=1))
    assert failed_first > 0.15, failed_first                      # the list rounds have work
    if spread >= 3.0:
        assert np.mean(~moved) > 0.5                              # most particles ran into the redraw cap: the current point


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("d,K,rounds,n", [(19, 3, 1, 3000), (32, 4, 3, 2999), (50, 2, 2, 1000), (100, 3, 4, 700), (33, 5, 24, 530)])
def test_blocked_rounds_with_several_modes_vs_oracle_and_multilane(dev, kernel, bc, d, K, rounds, n):
    """K > 1 at n_dim > 16 on the fast path (tempest/mcmc.py:225-249 applies per-cluster mu / L / Sigma^-1 at any dimension):
    the particles are grouped by mode into tiles of 16, every tile runs the matrix-core round kernel with ITS mode's factors,
    the failure lists of the rounds are kept per mode, and the particles still out of bounds go to the multi-lane kernel.
    Modes of very different sizes (one of them EMPTY, one smaller than a tile), an ensemble that is not a multiple of 16, a
    good share of first attempts out of bounds: proposals and both Mahalanobis forms equal the oracle's to rounding."""
    rs = np.random.RandomState(77 + d + K)
    means = 0.5 + 0.08 * rs.randn(K, d)
    covs = np.empty((K, d, d))
    for k in range(K):
        A = rs.randn(d, d) * ((0.03 + 0.02 * k) / np.sqrt(d))
        covs[k] = A @ A.T + (1e-4 + 5e-5 * k) * np.eye(d)
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([8.0, 1e6, 3.0, 25.0, 12.0][:K])
    sigmas = (np.array([0.6, 0.9, 0.4, 0.7, 0.5][:K])) * (2.38 / np.sqrt(d) if kernel == "rwm" else 1.0)
    probs = np.array([0.55, 0.0, 0.44, 0.01, 0.3][:K])          # mode 1 is empty; mode 3 (K >= 4) holds a handful of particles
    probs = probs / probs.sum()
    assign = rs.choice(K, size=n, p=probs).astype(np.int32)
    u = np.clip(means[assign] + 0.1 * rs.randn(n, d), 0.002, 0.998)
    u[: n // 3, rs.randint(d)] = 0.001                           # a third of the ensemble sits on a wall
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 4711, 9, 2_500_000_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    at = torch.from_numpy(assign).to(dev)
    got = {}
    for variant in (4, 3, 40):                           # 40: variant 4 with the list rounds NOT fanned out (TPH_OPT_BLK_FAN = 0)
        c = ctx_for(d)
        c.set_option(0, 4 if variant == 40 else variant)
        c.set_option(4, rounds if variant != 3 else 0)
        c.set_option(16, 1 if d in (19, 50) else 2)      # TPH_OPT_BLK_TRIES
        c.set_option(17, 0 if variant == 40 else 1)      # TPH_OPT_BLK_FAN (per mode: width and next attempt from the mode's own list)
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), at, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[variant] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy())
        c.close()
    np.testing.assert_allclose(got[40][0], want_up, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(got[4][0], got[40][0], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(np.any(got[4][0] != u, axis=1), np.any(got[40][0] != u, axis=1))
    np.testing.assert_allclose(got[4][0], want_up, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(got[4][0], got[3][0], rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((got[4][0][:, strict] >= 0) & (got[4][0][:, strict] <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(got[4][1], want_mu, rtol=1e-9)
        np.testing.assert_allclose(got[4][2], want_mup, rtol=1e-8, atol=1e-8)
    z0 = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, omc.bc_flags(d, list(range(d)), []), seed, tick, item0)[0]
    failed = np.mean(np.any(np.abs(z0 - want_up) > 1e-9, axis=1))
    assert failed > 0.03
    if rounds > 1 and bc is None and d in (19, 50):      # one attempt per round: the redraw probe is n / (n - first-attempt failures)
        np.testing.assert_allclose(got[4][3][8], 1.0 / (1.0 - failed), rtol=1e-12)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("mfma", [1, 0])
def test_blocked_kernel_deferred_update_and_step_control(dev, kernel, mfma):
    """The blocked path inside a chain: deferred Metropolis update (pending mask) and the carried Mahalanobis form give the
    same chain, bit for bit, as the blocked path with the in-place update (matrix-core and scalar-cache round kernels)."""
    rs = np.random.RandomState(5)
    d, n = 50, 4000
    means = 0.5 + 0.02 * rs.randn(1, d)
    A = rs.randn(d, d) * (0.03 / np.sqrt(d))
    covs = (A @ A.T + 1e-4 * np.eye(d))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([1e6]), dev)
    st = torch.from_numpy(np.array([0.5 * (2.38 / np.sqrt(d) if kernel == "rwm" else 1.0)])).to(dev)
    u0 = np.clip(means[0] + 0.05 * rs.randn(n, d), 0.002, 0.998)
    c = ctx_for(d)
    c.set_option(0, 4)
    c.set_option(15, mfma)

    def like(up):
        x = 20 * up - 10
        return -0.5 * (x * x).sum(dim=0) * 0.02

    def chain(deferred):
        u = soa(u0, dev)
        logl = like(u).clone()
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        sums = c.empty(2)
        pend = torch.zeros(n, dtype=torch.uint8, device=dev) if deferred else None
        out = []
        for step in range(3):
            tick = 30 + 2 * step
            c.propose(kernel, u, None, modes, st, None, 9, tick, 0, up, mu_, mup, pending=pend)
            lp = like(up)
            c.accept(kernel, 0.9, u, None, logl, up, None, lp, mu_, mup, None, 1, modes.dof_dev, 9, tick + 1, 0, sums, pending=pend)
            out.append((up.clone(), logl.clone(), sums.clone()))
        if deferred:
            c.propose(kernel, u, None, modes, st, None, 9, 999, 0, up, mu_, mup, pending=pend)
            assert int(pend.sum().item()) == 0
        return u, out
    ua, ta = chain(False)
    ub, tb = chain(True)
    for a, b in zip(ta, tb):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(ua, ub)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("d,lanes,zl", [(19, 0, 0), (33, 2, 0), (50, 0, 0), (50, 5, 0), (64, 3, 0),
                                        # zl = rows of z kept in LDS (TPH_OPT_SM_THRESHOLD; 0 = the default 32), the rest in
                                        # the lane's column of global scratch
                                        (50, 0, 16), (37, 3, 48), (50, 2, 64), (72, 0, 0), (100, 3, 0), (100, 0, 48), (100, 2, 112)])
def test_stage_machine_proposal_kernel_vs_oracle_and_multilane(dev, kernel, bc, d, lanes, zl):
    """TPH_OPT_PROPOSE_VARIANT 5 (propose_sm.hip, the row walker): the redraw loop of mcmc.py:239-249 with a lane per
    attempt, several attempts of a particle in flight, four rows at a time with early exit, the first in-bounds attempt in
    attempt order wins.  Same counter-based draws and row arithmetic as the other kernels: on an ensemble where most attempts
    leave the cube (tens of attempts per particle, some particles at the 256-attempt cap) the proposals and both
    Mahalanobis forms equal the oracle's sequential loop and the multi-lane kernel to rounding."""
    from tempest_amd.device import HipContext
    rs = np.random.RandomState(131 + d)
    n = 1500 if d < 64 else 1100            # not a multiple of the 16-particle queue chunks
    means = 0.5 + 0.05 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (0.29 ** 2 / 2.0))[None]       # as broad as the prior: the early iterations of a run
    _, chol, inv = ps.mode_statistics(means, covs)
    dof = np.array([1e6])
    sigmas = np.array([2.38 / np.sqrt(d)]) * (1.0 if kernel == "rwm" else 1.0)
    assign = np.zeros(n, dtype=np.int32)
    u = rs.rand(n, d)
    u[:40] = np.clip(u[:40], 0.45, 0.55)                             # a few easy particles: attempt 0 or 1 succeeds
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 991, 7, 3_000_000_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    capped = np.all(want_up == u, axis=1) if kernel == "rwm" else np.zeros(n, bool)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    got = {}
    for variant in (5, 3):
        c = HipContext(d, device=0)
        c.set_option(0, variant)
        c.set_option(10, lanes)            # TPH_OPT_SM_LANES
        c.set_option(11, zl)               # TPH_OPT_SM_THRESHOLD
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[variant] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy())
        c.close()
    np.testing.assert_allclose(got[5][0], want_up, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(got[5][0], got[3][0], rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((got[5][0][:, strict] >= 0) & (got[5][0][:, strict] <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(got[5][1], want_mu, rtol=1e-9)
        np.testing.assert_allclose(got[5][2], want_mup, rtol=1e-8, atol=1e-8)
    else:
        assert not got[5][1].any() and not got[5][2].any()
    # the regime really is the redraw one: many attempts per particle on average (the kernel's own probe, state[8])
    assert got[5][3][8] > (3.0 if bc is None else 2.0), got[5][3][8]
    _ = capped


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_stage_machine_redraw_cap_proposes_the_current_point(dev, kernel):
    """All 256 attempts out of bounds (a step size far too large: the RWM runaway of DESIGN section 9): the current point is
    proposed, like the other kernels do, and the probe reports the cap."""
    from tempest_amd.device import HipContext
    rs = np.random.RandomState(4)
    d, n = 40, 777
    means = np.full((1, d), 0.5)
    covs = (np.eye(d) * 0.08)[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([1e6]), dev)
    u = rs.rand(n, d)
    st = torch.from_numpy(np.array([0.99 if kernel == "tpcn" else 30.0])).to(dev)
    c = HipContext(d, device=0)
    c.set_option(0, 5)
    up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
    state = c.zeros(10)
    c.propose(kernel, soa(u, dev), None, modes, st, None, 5, 3, 0, up, mu_, mup, ctl=state)
    got = aos(up)
    if kernel == "rwm":
        np.testing.assert_array_equal(got, u)
        assert state.cpu().numpy()[8] == 256.0
    else:                                     # tpCN at sigma 0.99 contracts towards the mean: most attempts do succeed
        want = omc.propose(kernel, u, np.zeros(n, np.int32), means, chol, inv, np.array([1e6]), np.array([0.99]), omc.bc_flags(d), 5, 3, 0)[0]
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13)
    c.close()


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_stage_machine_kernel_deferred_update_and_step_control(dev, kernel):
    """The stage-machine path inside a chain: the deferred Metropolis update (pending mask resolved by its opening pass) and
    the carried Mahalanobis form give the same chain, bit for bit, as the in-place update; and a chain through it equals the
    chain through the multi-lane kernel to rounding in its first step."""
    rs = np.random.RandomState(15)
    d, n = 50, 3000
    means = 0.5 + 0.02 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (0.2 ** 2 / 2.0))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([1e6]), dev)
    st = torch.from_numpy(np.array([0.8 * 2.38 / np.sqrt(d)])).to(dev)
    u0 = rs.rand(n, d)
    c = ctx_for(d)
    c.set_option(0, 5)

    def like(up):
        x = 20 * up - 10
        return -0.5 * (x * x).sum(dim=0) * 0.02

    def chain(deferred):
        u = soa(u0, dev)
        logl = like(u).clone()
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        sums = c.empty(2)
        pend = torch.zeros(n, dtype=torch.uint8, device=dev) if deferred else None
        ctl = c.zeros(10)
        ctl[6] = 0.9                             # beta of the run (tph_accept reads it from the block)
        out = []
        for step in range(3):
            ctl[0] = float(step)                 # steps done: from the second step on the form at u is carried
            c.propose(kernel, u, None, modes, st, None, 9, 30, 0, up, mu_, mup, ctl=ctl, pending=pend)
            lp = like(up)
            c.accept(kernel, 0.9, u, None, logl, up, None, lp, mu_, mup, None, 1, modes.dof_dev, 9, 31, 0, sums, ctl=ctl, pending=pend)
            out.append((up.clone(), logl.clone(), sums.clone(), mu_.clone()))
        if deferred:
            c.propose(kernel, u, None, modes, st, None, 9, 999, 0, up, mu_, mup, pending=pend)
            assert int(pend.sum().item()) == 0
        return u, out
    ua, ta = chain(False)
    ub, tb = chain(True)
    for a, b in zip(ta, tb):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(ua, ub)
    assert 0 < float(ta[0][2][0]) < n
    c.set_option(0, 0)


@pytest.mark.parametrize("d", [3, 10, 50])
def test_gather_through_the_row_mirror_equals_the_dimension_major_gather(dev, d):
    """tph_gather from the lazily filled row-major mirror (TPH_OPT_ROW_MIRROR, default) against the dimension-major gather:
    the same rows bit for bit -- after appends (the mirror catches up with the new rows only), after a reload and a clear."""
    from tempest_amd.device import HipContext, OPT_ROW_MIRROR
    rs = np.random.RandomState(5)
    c = HipContext(d, 0)

    def append(n, beta):
        u = torch.from_numpy(rs.rand(d, n)).to(dev)
        c.history_append(u, 20 * u - 10, torch.from_numpy(rs.randn(n)).to(dev), beta, 0.0)

    def both(n_out):
        idx = torch.from_numpy(rs.randint(0, c.size, size=n_out)).to(dev)
        out = []
        for mode in (1, 0):
            c.set_option(OPT_ROW_MIRROR, mode)
            u, x, l = c.empty(d, n_out), c.empty(d, n_out), c.empty(n_out)
            c.gather(idx, u, x, l)
            out.append((u, x, l))
        c.set_option(OPT_ROW_MIRROR, 1)
        for a, b in zip(*out):
            assert torch.equal(a, b)
        hu = torch.from_numpy(c.history_read(0, 0, c.size, soa=True)).to(dev)         # [d][size]
        assert torch.equal(out[0][0], hu[:, idx])
    append(3000, 0.0)
    both(1000)
    append(70001, 0.1)                   # mirror grows and packs only the new rows
    both(4097)
    both(63)                             # small gather far behind nothing: still the mirror (it is up to date)
    append(129, 0.2)
    both(5000)
    n = 900
    u = rs.rand(n, d)
    c.history_load(u, 20 * u - 10, rs.randn(n), [0.0, 0.3], [0.0, -1.0], [400, 500])
    both(777)
    c.close()


@pytest.mark.parametrize("d", [16, 19, 32, 50, 77, 100])
def test_second_moments_on_the_matrix_cores_equal_the_vector_kernel(dev, d):
    """TPH_OPT_COV_KERNEL: the FP64-MFMA form of the centred second moments (n_dim >= 16) against the register-blocked VALU
    kernel and against NumPy: integer multiplicities (proposal fit) and real weights (mixture M-step), ragged row counts."""
    from tempest_amd.device import HipContext, OPT_COV_KERNEL
    rs = np.random.RandomState(d)
    n = 64 * 37 + 29
    c = HipContext(d, 0)
    A = rs.randn(d, d) * 0.05
    u = np.clip(0.5 + rs.randn(n, d) @ A.T + 0.1 * rs.randn(n, 1), 0.0, 1.0)
    c.history_load(u, 20 * u - 10, rs.randn(n), [0.0], [0.0], [n])
    counts = rs.poisson(np.exp(rs.randn(n) - 0.5)).astype(np.int32)
    ct = torch.from_numpy(counts).to(dev)
    got = {}
    for mode in (1, 2):
        c.set_option(OPT_COV_KERNEL, mode)
        means, covs, chol, inv, winv = c.fit_modes(ct, None, 1, n)
        got[mode] = covs.cpu().numpy()[0].reshape(d, d)
    np.testing.assert_allclose(got[2], got[1], rtol=1e-11, atol=1e-16)
    ref = np.cov(u.T, fweights=counts, ddof=0)
    scale = np.abs(ref).max()
    assert np.abs(got[2] - ref).max() / scale < 2.0 / counts.sum() + 1e-12        # student.py's normalisation differs by O(1/N)
    # real weights: the M-step entry point
    X = torch.from_numpy(np.ascontiguousarray(u.T)).to(dev)
    w = torch.from_numpy(rs.rand(n) * (rs.rand(n) < 0.7)).to(dev)
    mu = torch.from_numpy(u.mean(axis=0)).to(dev)
    out = {}
    for mode in (1, 2):
        c.set_option(OPT_COV_KERNEL, mode)
        out[mode] = c.x_weighted_cov(X, w, mu).cpu().numpy().reshape(d, d)
    np.testing.assert_allclose(out[2], out[1], rtol=1e-11, atol=1e-16)
    wn = w.cpu().numpy()
    xc = u - u.mean(axis=0)
    np.testing.assert_allclose(out[2], (xc * wn[:, None]).T @ xc, rtol=1e-10, atol=1e-14)
    assert np.allclose(out[2], out[2].T)
    c.close()


def test_fit_mvstud_drop_in_against_the_reference_outputs(dev):
    """tempest_amd.student.fit_mvstud (student.py:6-116, effective form F5) on the data sets whose reference outputs are in
    g7_modes.npz (Gaussian, t3, narrow 10-D, degenerate -> ridge), then the shapes the reference's tests/test_student.py
    feeds it: 1-D, 2-D, correlated, columns of very different scale, constant columns."""
    from tempest_amd.student import fit_mvstud
    g = np.load(os.path.join(G, "g7_modes.npz"))
    for k in range(int(g["n_fit"])):
        mu, Sig, nu = fit_mvstud(g[f"fit{k}_data"], tolerance=1e-8, max_iter=5)
        assert nu == np.inf == float(g[f"fit{k}_nu"])
        np.testing.assert_allclose(mu, g[f"fit{k}_mu"], rtol=1e-14, atol=0)
        ref = g[f"fit{k}_Sigma"]
        if np.min(np.diag(ref)) < 1e-25:
            # five constant columns: NumPy's covariance leaves rounding noise of 1e-31 there, which happens to pass its
            # Cholesky test; the device's centred moments are exactly zero, so the matrix is singular and the reference's own
            # rule (student.py:60-64) adds max(1e-6, 1e-6 |tr|) to the diagonal
            ref = ref + max(1e-6, 1e-6 * abs(np.trace(ref))) * np.eye(ref.shape[0])
        np.testing.assert_allclose(Sig, ref, rtol=1e-9, atol=1e-18)
    rs = np.random.RandomState(12)
    cases = [rs.randn(500, 1) * 2.0 + 3.0,
             rs.randn(300, 2),
             rs.multivariate_normal([0.0, 0.0, 0.0], [[1.0, 0.8, 0.3], [0.8, 1.0, 0.5], [0.3, 0.5, 1.0]], size=1000),
             rs.randn(400, 3) * np.array([1e-3, 1.0, 1e3]),
             np.tile(np.array([[1.0, 2.0, 3.0]]), (100, 1))]
    for data in cases:
        mu, Sig, nu = fit_mvstud(data)
        wmu, wSig, wnu = ps.fit_mvstud_effective(data)
        assert mu.shape == (data.shape[1],) and Sig.shape == (data.shape[1],) * 2 and nu == wnu
        np.testing.assert_allclose(mu, wmu, rtol=1e-14, atol=0)
        np.testing.assert_allclose(Sig, wSig, rtol=1e-9, atol=1e-16)
        np.linalg.cholesky(Sig)                                  # positive definite, as the reference guarantees
    np.testing.assert_array_equal(fit_mvstud(cases[2])[1], fit_mvstud(cases[2])[1])      # reproducible


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("d", [17, 97])
def test_stage_machine_odd_sizes_vs_multilane(dev, kernel, d):
    """The row walker at the edges of its index arithmetic: a dimension just above 16 and one just below 100 (neither a
    multiple of 4 nor of 16), ensembles smaller than a queue chunk, of exactly one tile, of one more; every (lanes per
    particle, rows of z in LDS) the options allow at these sizes.  Same proposals as the multi-lane kernel to rounding."""
    from tempest_amd.device import HipContext
    rs = np.random.RandomState(7 * d)
    means = 0.5 + 0.05 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (0.25 ** 2 / 2.0))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([5.0]), dev)
    st = torch.from_numpy(np.array([2.38 / np.sqrt(d)])).to(dev)
    for n, lanes, zl in ((1, 0, 0), (3, 6, 16), (5, 1, 0), (64, 0, 48), (65, 3, 0), (257, 4, 112)):
        u = rs.rand(n, d)
        got = {}
        for variant in (5, 3):
            c = HipContext(d, device=0)
            c.set_option(0, variant)
            c.set_option(10, lanes)
            c.set_option(11, zl)
            up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
            c.propose(kernel, soa(u, dev), None, modes, st, None, 4242, 9, 123456789, up, mu_, mup)
            got[variant] = (aos(up), mup.cpu().numpy())
            c.close()
        np.testing.assert_allclose(got[5][0], got[3][0], rtol=1e-11, atol=1e-13, err_msg=f"n={n}")
        assert np.all((got[5][0] >= 0) & (got[5][0] <= 1))
        if kernel == "tpcn":
            np.testing.assert_allclose(got[5][1], got[3][1], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("d,rounds,bc", [(50, 1, None), (100, 3, None), (65, 2, "mixed"), (33, 4, None)])
def test_blocked_rounds_with_panels_staged_in_lds_equal_the_streamed_ones(dev, kernel, d, rounds, bc):
    """TPH_OPT_BLK_STAGE: the matrix-core rounds with every panel's blocks of L and L^-1 staged in LDS
    once per workgroup issue the SAME matrix instructions on the same operands as the kernel that streams the blocks to each wave:
    proposals, both Mahalanobis forms, the failure counts of the rounds and the redraw probe are bit for bit the same (full
    ensemble, list rounds with fan-out, stragglers through the screened launch)."""
    rs = np.random.RandomState(50 + d)
    n = 3000
    means = 0.5 + 0.03 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (0.10 ** 2 / 2.0))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    u = np.clip(0.5 + 0.22 * rs.randn(n, d), 0.01, 0.99)
    dof, sigmas = np.array([1e6]), np.array([2.38 / np.sqrt(d)])
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    seed, tick, item0 = 77, 3, 1_000_000
    got = {}
    for stage in (0, 1):
        c = ctx_for(d)
        c.set_option(0, 4)                 # blocked path
        c.set_option(4, rounds)
        c.set_option(15, 1)                # matrix cores
        c.set_option(16, 1)                # one try per round (the staged kernel's case)
        c.set_option(20, stage)            # TPH_OPT_BLK_STAGE
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[stage] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy())
        c.set_option(20, 0); c.set_option(4, 0); c.set_option(16, 0); c.set_option(0, 0)
    for a, b in zip(got[0], got[1]):
        np.testing.assert_array_equal(a, b)
    want_up = omc.propose(kernel, u, np.zeros(n, dtype=np.int32), means, chol, inv, dof, sigmas, flags, seed, tick, item0)[0]
    np.testing.assert_allclose(got[1][0], want_up, rtol=1e-11, atol=1e-13)
    assert np.any(np.any(got[1][0] != u, axis=1))
