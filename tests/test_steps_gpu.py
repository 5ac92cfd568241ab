"""Step plugins on the device against golden vectors of the reference's steps (run on the GPU box)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _state_from_history(u, logl, beta_t, logz_t, n_t):
    from tempest_amd.state_manager import StateManager
    d = u.shape[1]
    st = StateManager(d)
    off = np.concatenate([[0], np.cumsum(n_t)])
    hist = {"u": [u[off[t]:off[t + 1]] for t in range(len(n_t))],
            "x": [20 * u[off[t]:off[t + 1]] - 10 for t in range(len(n_t))],
            "logl": [logl[off[t]:off[t + 1]] for t in range(len(n_t))],
            "beta": list(beta_t), "logz": list(logz_t), "iter": list(range(len(n_t)))}
    st.update_from_dict({"_history": hist, "n_dim": d})
    return st


def test_reweighter_decisions_match_reference():
    """steps/reweight.py:341-495 on frozen histories: same beta (bitwise), ESS, logZ, cv, weights and -- in ESS
    mode -- the same sequence of trial betas as the reference evaluated (incl. the two regressions pinned by the
    reference's tests/test_steps.py:100-200)."""
    from tempest_amd import config as C
    from tempest_amd.steps import Reweighter
    g = np.load(os.path.join(G, "g3_reweighter.npz"))
    for k in range(int(g["n_cases"])):
      for depth in (1, 4):          # one beta per pass (as the reference) and exact-bisection batching: same decisions
        n_p, er, vv, bp = g[f"c{k}_cfg"]
        st = _state_from_history(g[f"c{k}_u"], g[f"c{k}_logl"], g[f"c{k}_beta_t"], g[f"c{k}_logz_t"], g[f"c{k}_n_t"])
        assert st.get_history_length() == len(g[f"c{k}_beta_t"])
        st.update_current({"beta": float(bp), "iter": st.get_history_length(), "logz": 0.0, "calls": 0})
        rw = Reweighter(state=st, pbar=None, n_particles=int(n_p), ess_ratio=float(er),
                        volume_variation=None if vv < 0 else float(vv), ESS_TOLERANCE=C.ESS_TOLERANCE,
                        BETA_TOLERANCE=C.BETA_TOLERANCE, BETA_RTOL=C.BETA_RTOL, METRIC_ATOL=C.METRIC_ATOL,
                        METRIC_ATOL_CV=C.METRIC_ATOL_CV)
        rw.batch_depth = depth
        trace = []
        orig = rw._eval

        def traced(beta, _o=orig, _t=trace):
            if beta not in _t:
                _t.append(beta)
            return _o(beta)
        rw._eval = traced
        w = rw.run()
        ref = g[f"c{k}_out"]
        tag = str(g[f"c{k}_tag"])
        assert st.get_current("beta") == ref[0], (k, tag, st.get_current("beta"), ref[0])
        np.testing.assert_allclose([st.get_current("ess"), st.get_current("logz"), st.get_current("cv")], ref[1:],
                                   rtol=1e-9, err_msg=tag)
        np.testing.assert_allclose(np.asarray(w), g[f"c{k}_weights"], rtol=1e-9, atol=1e-300)
        assert abs(np.sum(w) - 1.0) < 1e-12 and len(w) == len(g[f"c{k}_logl"])
        assert st.get_current("iter") == len(g[f"c{k}_beta_t"]) + 1
        if vv < 0:
            ref_trace = [b for i, b in enumerate(g[f"c{k}_trace"]) if b not in g[f"c{k}_trace"][:i]]
            assert trace == ref_trace, (k, tag)
            if depth == 4 and len(ref_trace) > 6:
                assert rw.n_passes < len(ref_trace) / 2          # several bisection levels per pass


def test_reweighter_first_iteration_contract():
    """reference tests/test_steps.py:39-56."""
    from tempest_amd.state_manager import StateManager
    from tempest_amd.steps import Reweighter
    st = StateManager(2)
    st.update_current({"iter": 0, "beta": 0.0, "logz": 0.0, "calls": 0})
    w = Reweighter(state=st, pbar=None, n_particles=32, ess_ratio=2.0).run()
    assert len(w) == 32
    np.testing.assert_allclose(w, 1.0 / 32)
    assert st.get_current("beta") == 0.0 and st.get_current("logz") == 0.0
    assert st.get_current("ess") == 64 and st.get_current("iter") == 1


def test_state_manager_interface():
    """Key validation, copy semantics, history shapes, compute_logw_and_logz vs the golden vectors
    (state_manager.py:178-480; reference tests/test_state_manager.py)."""
    from tempest_amd.state_manager import StateManager
    g = np.load(os.path.join(G, "g1_logw.npz"))
    k = 4
    logl, bt, zt, nt = g[f"c{k}_logl"], g[f"c{k}_beta_t"], g[f"c{k}_logz_t"], g[f"c{k}_n_t"]
    st = StateManager(3)
    assert st.compute_logw_and_logz(1.0)[1] == -np.inf and st.get_history_length() == 0
    assert st.get_last_history("beta", default=7) == 7
    rs = np.random.RandomState(0)
    off = np.concatenate([[0], np.cumsum(nt)])
    for t in range(len(nt)):
        u = rs.rand(int(nt[t]), 3)
        st.update_current({"u": u, "x": 20 * u - 10, "logl": logl[off[t]:off[t + 1]], "beta": float(bt[t]),
                           "logz": float(zt[t]), "iter": t})
        got = st.get_current("u")
        got[:] = 0
        np.testing.assert_array_equal(st.get_current("u"), u)          # getters hand out copies
        st.commit_current_to_history(strict=True)
    assert st.get_history_length() == len(nt)
    assert st.get_history("u").shape == (len(nt), int(nt[0]), 3)
    assert st.get_history("x", flat=True).shape == (int(nt.sum()), 3)
    np.testing.assert_array_equal(st.get_history("logl", index=2), logl[off[2]:off[3]])
    np.testing.assert_array_equal(st.get_last_history("u"), u)
    for j in range(4):
        bf = float(g[f"c{k}_b{j}_beta"])
        lw, lz = st.compute_logw_and_logz(bf)
        np.testing.assert_allclose(lw, g[f"c{k}_b{j}_logw"], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(lz, g[f"c{k}_b{j}_logz"], rtol=1e-11)
        lwu, _ = st.compute_logw_and_logz(bf, normalize=False)
        np.testing.assert_allclose(lwu, g[f"c{k}_b{j}_logw_unnorm"], rtol=1e-10, atol=1e-9)
    with pytest.raises(ValueError):
        st.set_current("bogus", 1)
    with pytest.raises(ValueError):
        st.get_history("bogus")
    with pytest.raises(IndexError):
        st.get_history("logl", index=99)
    st.set_current("logl", None)
    with pytest.raises(ValueError):
        st.commit_current_to_history(strict=True)
    d = st.to_dict()
    st2 = StateManager.from_dict(d)
    np.testing.assert_array_equal(st2.get_history("x", flat=True), st.get_history("x", flat=True))
    np.testing.assert_allclose(st2.compute_logw_and_logz(0.3)[1], st.compute_logw_and_logz(0.3)[1], rtol=1e-14)


def test_tools_wrappers_match_reference_values():
    from tempest_amd import tools
    g = np.load(os.path.join(G, "g2_tools.npz"))
    assert tools.effective_sample_size(np.ones(4)) == pytest.approx(4.0)          # reference tests/test_tools.py
    assert tools.increment_logz(np.zeros(100)) == pytest.approx(np.log(100))
    for k in range(4):
        assert tools.effective_sample_size(g[f"ess{k}_w"]) == pytest.approx(float(g[f"ess{k}"]), rel=1e-12)
        idx, wt = tools.trim_weights(np.arange(g[f"ess{k}_w"].size), g[f"ess{k}_w"].copy(), 0.99, 1000)
        np.testing.assert_array_equal(idx, g[f"trim{k}_0_idx"])
        np.testing.assert_allclose(wt, g[f"trim{k}_0_w"], rtol=1e-12)
    np.testing.assert_allclose(tools.increment_logz(g["inc_logz_in"]), g["inc_logz_out"], rtol=1e-12)
    np.testing.assert_allclose(tools.volume_variation(g["vv0_x"], g["vv0_w"]), g["vv0"], rtol=1e-8)
    np.testing.assert_allclose(tools.volume_variation(g["vv1_x"]), g["vv1"], rtol=1e-8)
    np.testing.assert_allclose(tools.volume_variation(g["vv2_x"], g["vv2_w"]), g["vv2"], rtol=1e-5)   # rank-deficient + ridge
    assert tools.volume_variation(g["vv3_x"]) == 1e10
    g4 = np.load(os.path.join(G, "g4_resample.npz"))
    np.random.seed(0)
    idx = tools.systematic_resample(int(g4["c2_size"]), g4["c2_w"], random_state=0)
    np.testing.assert_array_equal(idx, g4["c2_idx"])


def test_mode_statistics_from_arrays_and_parallel_mcmc():
    """ModeStatistics (modes.py:58-119) chol*chol^T = cov, cov*inv = I (reference tests/test_modes.py:75-89) and the
    host-array drop-in `parallel_mcmc` (mcmc.py:414-508)."""
    from tempest_amd.mcmc import parallel_mcmc
    from tempest_amd.modes import ModeStatistics
    rs = np.random.RandomState(1)
    d, n = 4, 512
    A = rs.randn(d, d) * 0.05
    cov = A @ A.T + 1e-3 * np.eye(d)
    ms = ModeStatistics(np.full((1, d), 0.5), cov[None], np.array([1e6]))
    assert ms.K == 1 and ms.n_dim == d
    np.testing.assert_allclose(ms.chol_covariances[0] @ ms.chol_covariances[0].T, cov, rtol=1e-10)
    np.testing.assert_allclose(ms.covariances[0] @ ms.inv_covariances[0], np.eye(d), atol=1e-8)
    u = np.clip(0.5 + 0.03 * rs.randn(n, d), 0, 1)
    x = 20 * u - 10
    logl = -0.5 * np.sum(x ** 2, axis=1)
    np.random.seed(3)
    out = parallel_mcmc(u, x, logl, None, np.zeros(n, dtype=int), 1.0, ms, lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None),
                        lambda uu: 20 * uu - 10, None, n_steps=2, n_max=10, sample="tpcn")
    u2, x2, l2, blobs, eff, acc, steps, calls = out
    assert u2.shape == (n, d) and blobs is None and calls == steps * n and 2 * d <= steps <= 10 * d
    np.testing.assert_allclose(x2, 20 * u2 - 10, rtol=1e-14)
    np.testing.assert_allclose(l2, -0.5 * np.sum(x2 ** 2, axis=1), rtol=1e-12)
    assert 0 < acc <= 1 and eff > 0 and np.mean(np.any(u2 != u, axis=1)) > 0.5
    ms2 = ModeStatistics.from_global(u, np.ones(n))
    np.testing.assert_allclose(ms2.means[0], np.median(u, axis=0), atol=0.01)
    assert ms2.degrees_of_freedom[0] == 1e6


def test_named_mcmc_wrappers_equal_parallel_mcmc():
    """parallel_t_preconditioned_crank_nicolson / parallel_random_walk_metropolis (mcmc.py:511-676) are parallel_mcmc with the
    kernel fixed: same seed, same chain."""
    from tempest_amd import mcmc
    from tempest_amd.modes import ModeStatistics
    rs = np.random.RandomState(4)
    d, n = 3, 256
    u = np.clip(0.5 + 0.05 * rs.randn(n, d), 0, 1)
    x = 20 * u - 10
    logl = -0.5 * np.sum(x ** 2, axis=1)
    ms = ModeStatistics.from_global(u, np.ones(n), seed=2)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)      # noqa: E731
    prior = lambda uu: 20 * uu - 10                                # noqa: E731
    args = (u, x, logl, None, np.zeros(n, dtype=int), 0.7, ms, like, prior)
    for wrapper, name in ((mcmc.parallel_t_preconditioned_crank_nicolson, "tpcn"), (mcmc.parallel_random_walk_metropolis, "rwm")):
        np.random.seed(11)
        a = wrapper(*args, n_steps=2, n_max=20, verbose=False)
        np.random.seed(11)
        b = mcmc.parallel_mcmc(*args, n_steps=2, n_max=20, sample=name, verbose=False)
        for va, vb in zip(a, b):
            if va is None:
                assert vb is None
            else:
                np.testing.assert_array_equal(np.asarray(va), np.asarray(vb))
        assert a[0].shape == (n, d) and a[6] >= 2 * d
