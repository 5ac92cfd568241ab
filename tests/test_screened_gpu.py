"""The screened-batch proposal kernel (csrc/propose_mf.hip, TPH_OPT_PROPOSE_VARIANT 6) through the C ABI.

The redraw loop of tempest/mcmc.py:239-249 with the attempts screened in low precision on the matrix cores and only the
survivors evaluated in FP64.  The claims pinned here:
  * the proposal equals the oracle's sequential loop to rounding and the FP64 row walker's (variant 5) BIT FOR BIT -- the
    screen may only ever remove attempts that fail in FP64 too;
  * with TPH_OPT_MF_AUDIT every attempt the screen removed is evaluated in FP64 as well: zero contradictions;
  * the FP32 Box-Muller pair of the screen stays within its budget (2^-13) of the FP64 pair of the same Philox block.
Run on the GPU box:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mcmc as omc  # noqa: E402
from oracle import ps  # noqa: E402

from tests.test_kernels_gpu import _Modes, aos, ctx_for, dev, soa  # noqa: E402,F401

OPT_VARIANT, OPT_SM_LANES, OPT_MF_LANES, OPT_MF_AUDIT = 0, 10, 13, 14


def _ensemble(d, n, seed, spread=0.29, easy=40):
    rs = np.random.RandomState(seed)
    means = 0.5 + 0.05 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (spread ** 2 / 2.0))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    u = rs.rand(n, d)
    u[:easy] = np.clip(u[:easy], 0.45, 0.55)          # a few easy particles: attempt 0 or 1 succeeds
    return means, chol, inv, u


def _run(variant, d, kernel, u, modes, st, ft, seed, tick, item0, dev, lanes=0, audit=False, sm_lanes=0):
    from tempest_amd.device import HipContext
    n = u.shape[0]
    c = HipContext(d, device=0)
    c.set_option(OPT_VARIANT, variant)
    c.set_option(OPT_MF_LANES, lanes)
    c.set_option(OPT_SM_LANES, sm_lanes)
    c.set_option(OPT_MF_AUDIT, 1 if audit else 0)
    up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
    state = c.zeros(10)
    c.propose(kernel, soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
    out = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy(), c.mf_counters() if variant == 6 else None)
    c.close()
    return out


def test_screen_normals_stay_within_their_error_budget(dev):
    """|z~ - z| of the screen's FP32 Box-Muller pair against tph_rng::normal2 on the same Philox blocks: 2^28 blocks of the
    stream the proposals use plus hand-made blocks at the ends of u1 and around the switch of the logarithm.  The budget in
    the row error tables is 2^-13 = 1.22e-4; the measured maximum has to leave half of it."""
    c = ctx_for(24)
    err, zmax, cnt = c.mf_normals_error(seed=20260101, first=0, n_blocks=1 << 28)
    assert cnt == float(1 << 28)
    assert zmax > 5.0                                   # the sample does reach the tails
    assert err < 6.1e-5, err
    err_e, _, _ = c.mf_normals_error(seed=1, first=0, n_blocks=1 << 22, edge=True)
    assert err_e < 6.1e-5, err_e
    print(f"screen normals: max |z~ - z| = {err:.3e} over 2^28 blocks (max |z| {zmax:.2f}), {err_e:.3e} on the edge blocks")
    c.close()


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("d,lanes", [(19, 0), (33, 3), (50, 0), (50, 5), (64, 3), (65, 4), (100, 0), (100, 6), (112, 3)])
def test_screened_kernel_vs_oracle_and_row_walker(dev, kernel, bc, d, lanes):
    """On an ensemble where most attempts leave the cube (tens of attempts per particle, some particles at the 256-attempt
    cap): proposals and both Mahalanobis forms equal the oracle's sequential loop to rounding, the proposals equal the FP64
    row walker's bit for bit (n_dim <= 100, the walker's range), and the audit finds no attempt that the screen removed and
    FP64 would have kept."""
    n = 1500 if d < 64 else 1100
    means, chol, inv, u = _ensemble(d, n, 131 + d)
    dof = np.array([1e6])
    sigmas = np.array([2.38 / np.sqrt(d)])
    assign = np.zeros(n, dtype=np.int32)
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 991, 7, 3_000_000_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    got = _run(6, d, kernel, u, modes, st, ft, seed, tick, item0, dev, lanes=lanes, audit=True)
    np.testing.assert_allclose(got[0], want_up, rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((got[0][:, strict] >= 0) & (got[0][:, strict] <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(got[1], want_mu, rtol=1e-9)
        np.testing.assert_allclose(got[2], want_mup, rtol=1e-8, atol=1e-8)
    else:
        assert not got[1].any() and not got[2].any()
    cnt = got[4]
    assert cnt["contradictions"] == 0, cnt
    assert cnt["particles"] == n
    assert cnt["screened"] >= cnt["attempts"] - 256 * n       # every counted attempt went through the screen (capped ones: 256)
    if kernel == "rwm":                                        # (a capped tpCN proposal (u - mu) + mu may differ from u in its last bit)
        assert cnt["verified"] >= np.count_nonzero(np.any(got[0] != u, axis=1))
    assert got[3][8] > (3.0 if bc is None else 2.0), got[3][8]  # the regime is the redraw one (the kernel's own probe)
    assert got[3][8] == cnt["attempts"] / n
    if d <= 100:
        walk = _run(5, d, kernel, u, modes, st, ft, seed, tick, item0, dev)
        np.testing.assert_array_equal(got[0], walk[0])
        np.testing.assert_array_equal(got[2], walk[2])
        assert got[3][8] == walk[3][8]                          # the same attempts won
    # the product path (audit off) is the same kernel without the extra evaluations: same proposals, counters and probe
    fast = _run(6, d, kernel, u, modes, st, ft, seed, tick, item0, dev, lanes=lanes, audit=False)
    np.testing.assert_array_equal(fast[0], got[0])
    np.testing.assert_array_equal(fast[2], got[2])
    assert fast[3][8] == got[3][8] and fast[4]["attempts"] == got[4]["attempts"] and fast[4]["contradictions"] == 0


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_screened_kernel_redraw_cap_proposes_the_current_point(dev, kernel):
    """All 256 attempts out of bounds (a step size far too large: the RWM runaway of DESIGN section 9): the current point is
    proposed, like the other kernels do, and the probe reports the cap."""
    rs = np.random.RandomState(4)
    d, n = 40, 777
    means = np.full((1, d), 0.5)
    covs = (np.eye(d) * 0.08)[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([1e6]), dev)
    u = rs.rand(n, d)
    st = torch.from_numpy(np.array([0.99 if kernel == "tpcn" else 30.0])).to(dev)
    got = _run(6, d, kernel, u, modes, st, None, 5, 3, 0, dev, audit=True)
    assert got[4]["contradictions"] == 0
    if kernel == "rwm":
        np.testing.assert_array_equal(got[0], u)
        assert got[3][8] == 256.0
        assert got[4]["verified"] == 0                     # nothing survived the screen
    else:
        want = omc.propose(kernel, u, np.zeros(n, np.int32), means, chol, inv, np.array([1e6]), np.array([0.99]), omc.bc_flags(d), 5, 3, 0)[0]
        np.testing.assert_allclose(got[0], want, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_screened_kernel_deferred_update_and_step_control(dev, kernel):
    """The screened path inside a chain: the deferred Metropolis update (pending mask resolved by its opening pass) and the
    carried Mahalanobis form give the same chain, bit for bit, as the in-place update -- and the same chain, bit for bit, as
    the FP64 row walker."""
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

    def like(up):
        x = 20 * up - 10
        return -0.5 * (x * x).sum(dim=0) * 0.02

    def chain(deferred, variant):
        c.set_option(OPT_VARIANT, variant)
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
    ua, ta = chain(False, 6)
    ub, tb = chain(True, 6)
    uc, tc = chain(True, 5)
    for a, b, w in zip(ta, tb, tc):
        for x, y, z in zip(a, b, w):
            assert torch.equal(x, y)
            assert torch.equal(x, z)
    assert torch.equal(ua, ub) and torch.equal(ua, uc)
    assert 0 < float(ta[0][2][0]) < n
    c.set_option(OPT_VARIANT, 0)
    c.close()


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("d", [17, 97, 111])
def test_screened_kernel_odd_sizes_vs_multilane(dev, kernel, d):
    """The edges of the index arithmetic: dimensions that are neither a multiple of 2, 4 nor 16; ensembles smaller than a
    queue chunk, of exactly one window set, of one more; every window length.  Same proposals as the multi-lane kernel to
    rounding, no contradiction in the audit."""
    rs = np.random.RandomState(7 * d)
    means = 0.5 + 0.05 * rs.randn(1, d)
    A = rs.randn(d, d) / np.sqrt(d)
    covs = ((A @ A.T + np.eye(d)) * (0.25 ** 2 / 2.0))[None]
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([5.0]), dev)
    st = torch.from_numpy(np.array([2.38 / np.sqrt(d)])).to(dev)
    for n, lanes in ((1, 0), (3, 6), (5, 3), (16, 3), (17, 3), (64, 0), (65, 4), (257, 5)):
        u = rs.rand(n, d)
        a = _run(6, d, kernel, u, modes, st, None, 4242, 9, 123456789, dev, lanes=lanes, audit=True)
        b = _run(3, d, kernel, u, modes, st, None, 4242, 9, 123456789, dev)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-13, err_msg=f"n={n}")
        assert np.all((a[0] >= 0) & (a[0] <= 1))
        assert a[4]["contradictions"] == 0 and a[4]["particles"] == n
        if kernel == "tpcn":
            np.testing.assert_allclose(a[2], b[2], rtol=1e-8, atol=1e-8)


def test_screened_kernel_badly_scaled_factor(dev):
    """A Cholesky factor whose entries span twelve orders of magnitude (a funnel-like target late in a run: one very narrow
    and one prior-wide direction, strongly coupled) -- entries below FP16's range after scaling are charged their full size in
    the row error tables, so the screen stays conservative: audit clean, proposals equal to the row walker's bit for bit."""
    d, n = 48, 2000
    rs = np.random.RandomState(99)
    scales = np.logspace(-9, -0.6, d)[rs.permutation(d)]
    A = rs.randn(d, d) / np.sqrt(d)
    C = (A @ A.T + np.eye(d))
    covs = (C * np.outer(scales, scales))[None]
    means = np.full((1, d), 0.5)
    _, chol, inv = ps.mode_statistics(means, covs)
    modes = _Modes(means, chol, inv, np.array([1e6]), dev)
    st = torch.from_numpy(np.array([2.38 / np.sqrt(d)])).to(dev)
    u = np.clip(0.5 + (rs.randn(n, d) * scales), 1e-6, 1 - 1e-6)
    u[:, np.argmax(scales)] = rs.rand(n)                  # the wide direction really reaches the walls
    for kernel in ("tpcn", "rwm"):
        a = _run(6, d, kernel, u, modes, st, None, 77, 5, 0, dev, audit=True)
        b = _run(5, d, kernel, u, modes, st, None, 77, 5, 0, dev)
        assert a[4]["contradictions"] == 0
        np.testing.assert_array_equal(a[0], b[0])
        assert a[3][8] > 1.02                            # some first attempts do leave the cube


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("d,K,bc", [(24, 3, None), (50, 2, "mixed"), (100, 4, None), (33, 5, None)])
def test_screened_kernel_with_several_modes_vs_oracle_and_single_mode_launches(dev, kernel, d, K, bc):
    """tempest/mcmc.py:225-249 applies each walker's own cluster mean / factor / sigma in its redraw loop at any dimension.
    Several proposal modes on the screened batches: the modes are served one after the other over the
    particles of each (csrc/propose_mf.hip: tph_propose_mf_modes).  On a redraw-dominated ensemble with K modes of different
    means, factors, sigmas and sizes (one of them nearly empty): proposals and both Mahalanobis forms equal the oracle's
    sequential loop to rounding, every particle's proposal equals -- BIT FOR BIT -- what the one-mode kernel gives when run on
    that mode's particles alone with that mode's statistics, the audit finds no contradiction, and the probe is the mean number
    of attempts over ALL particles."""
    n = 1400 if d < 64 else 1000
    rs = np.random.RandomState(977 + d + K)
    means = 0.5 + 0.08 * rs.randn(K, d)
    covs = np.empty((K, d, d))
    for m in range(K):
        A = rs.randn(d, d) / np.sqrt(d)
        covs[m] = (A @ A.T + np.eye(d)) * ((0.22 + 0.04 * m) ** 2 / 2.0)
    _, chol, inv = ps.mode_statistics(means, covs)
    u = rs.rand(n, d)
    u[:30] = np.clip(u[:30], 0.45, 0.55)
    assign = rs.randint(0, K, size=n).astype(np.int32)
    assign[assign == K - 1] = 0                              # the last mode keeps three particles only
    assign[[5, 600, n - 1]] = K - 1
    dof = np.full(K, 1e6)
    sigmas = (2.38 / np.sqrt(d)) * (1.0 - 0.1 * np.arange(K))
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 4242, 11, 2_000_000_000
    want_up, want_mu, want_mup = omc.propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    from tempest_amd.device import HipContext
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    at = torch.from_numpy(assign).to(dev)
    got = {}
    for audit in (True, False):
        c = HipContext(d, device=0)
        c.set_option(OPT_VARIANT, 6)
        c.set_option(OPT_MF_AUDIT, 1 if audit else 0)
        up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
        state = c.zeros(10)
        c.propose(kernel, soa(u, dev), at, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
        got[audit] = (aos(up), mu_.cpu().numpy(), mup.cpu().numpy(), state.cpu().numpy(), c.mf_counters())
        c.close()
    g = got[True]
    np.testing.assert_allclose(g[0], want_up, rtol=1e-11, atol=1e-13)
    strict = np.nonzero(flags == 0)[0]
    assert np.all((g[0][:, strict] >= 0) & (g[0][:, strict] <= 1))
    if kernel == "tpcn":
        np.testing.assert_allclose(g[1], want_mu, rtol=1e-9)
        np.testing.assert_allclose(g[2], want_mup, rtol=1e-8, atol=1e-8)
    cnt = g[4]
    assert cnt["contradictions"] == 0 and cnt["particles"] == n, cnt
    assert g[3][8] == cnt["attempts"] / n and g[3][8] > 2.0, (g[3][8], cnt)
    np.testing.assert_array_equal(got[False][0], g[0])         # the product path (audit off): the same proposals and forms
    np.testing.assert_array_equal(got[False][2], g[2])
    assert got[False][4]["attempts"] == cnt["attempts"]
    # mode by mode against the ONE-mode kernel on that mode's particles (item0 shifted so that every particle keeps its draws)
    for m in range(K):
        rows = np.nonzero(assign == m)[0]
        one = _Modes(means[m:m + 1], chol[m:m + 1], inv[m:m + 1], dof[m:m + 1], dev)
        sm = torch.from_numpy(sigmas[m:m + 1].copy()).to(dev)
        for r in rows[:: max(1, len(rows) // 40)]:              # a sample of the mode's particles, each as an ensemble of one
            single = _run(6, d, kernel, u[r:r + 1], one, sm, ft, seed, tick, item0 + int(r), dev)
            np.testing.assert_array_equal(single[0][0], g[0][r])
            assert single[2][0] == g[2][r]


@pytest.mark.parametrize("bc", [None, "mixed"])
@pytest.mark.parametrize("d", [19, 33, 50, 65, 100, 112])
def test_forms_behind_the_screened_batches_on_the_matrix_cores(dev, bc, d):
    """TPH_OPT_FORMS_MFMA: the forms |L^-1 (u' - mu)|^2 of a screened step from the matrix-core blocks of the blocked rounds
    (tph_blkm_forms) instead of the lane-per-particle pass: the proposals are untouched (bit for bit), the forms equal the oracle's
    and the lane-per-particle pass's to rounding, the regime probe is the same number; also with a straggler list behind a
    blocked round (variant 4: the closing pass over the listed particles)."""
    from tempest_amd.device import HipContext
    n = 1300
    means, chol, inv, u = _ensemble(d, n, 77 + d)
    dof, sigmas = np.array([1e6]), np.array([2.38 / np.sqrt(d)])
    flags = omc.bc_flags(d, [1], [min(4, d - 1)]) if bc else omc.bc_flags(d)
    seed, tick, item0 = 313, 5, 9_000_000
    want_up, _, want_mup = omc.propose("tpcn", u, np.zeros(n, dtype=np.int32), means, chol, inv, dof, sigmas, flags, seed, tick, item0)
    modes = _Modes(means, chol, inv, dof, dev)
    st, ft = torch.from_numpy(sigmas).to(dev), torch.from_numpy(flags).to(dev)
    for variant, rounds in ((6, 0), (4, 2)):
        got = {}
        for mf in (0, 1):
            c = HipContext(d, device=0)
            c.set_option(OPT_VARIANT, variant)
            c.set_option(4, rounds)
            c.set_option(22, mf)                      # TPH_OPT_FORMS_MFMA
            up, mu_, mup = c.empty(d, n), c.empty(n), c.empty(n)
            state = c.zeros(10)
            c.propose("tpcn", soa(u, dev), None, modes, st, ft, seed, tick, item0, up, mu_, mup, ctl=state)
            got[mf] = (aos(up), mup.cpu().numpy(), state.cpu().numpy())
            c.close()
        np.testing.assert_array_equal(got[1][0], got[0][0])
        np.testing.assert_allclose(got[1][0], want_up, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(got[1][1], got[0][1], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(got[1][1], want_mup, rtol=1e-8, atol=1e-8)
        assert got[1][2][8] == got[0][2][8]
