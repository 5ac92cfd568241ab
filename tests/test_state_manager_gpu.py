"""The StateManager interface, behaviour by behaviour as the reference's tests/test_state_manager.py exercises it
(tempest/state_manager.py:178-666): key validation, copy semantics of every getter and setter, history views, the accessors
for the last entry and the length, strict commits, dict / file round trips and the results dict -- here on the
device-resident implementation (current arrays and history in HBM, host copies handed out)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sm(d=3):
    from tempest_amd.state_manager import StateManager
    return StateManager(d)


def _fill(st, n_it=3, n=10, seed=0):
    """n_it committed iterations of n particles; returns the host arrays that went in."""
    rs = np.random.RandomState(seed)
    d = st.n_dim
    ins = []
    for t in range(n_it):
        u = rs.rand(n, d)
        x = 20 * u - 10
        logl = -0.5 * np.sum(x ** 2, axis=1)
        st.update_current({"u": u, "x": x, "logl": logl, "beta": 0.1 * t, "logz": -1.0 - t, "iter": t, "calls": 100 * (t + 1),
                           "steps": 5 + t, "efficiency": 0.8, "ess": 0.9 * n, "acceptance": 0.3 + 0.1 * t})
        st.commit_current_to_history(strict=True)
        ins.append((u, x, logl))
    return ins


def test_initial_state_and_key_validation():
    from tempest_amd.state_manager import CURRENT_STATE_KEYS, HISTORY_STATE_KEYS
    st = _sm(3)
    assert st.n_dim == 3 and st._results_dict is None and st.get_history_length() == 0
    cur = st.get_current()
    assert set(cur) == set(CURRENT_STATE_KEYS) and all(v is None for v in cur.values())
    for key in HISTORY_STATE_KEYS - {"u", "x", "logl"}:
        assert len(st.get_history(key)) == 0
    for call in (lambda: st.set_current("nope", 1), lambda: st.update_current({"beta": 0.1, "nope": 2}),
                 lambda: st.get_current("nope"), lambda: st.get_history("nope"), lambda: st.get_last_history("nope")):
        with pytest.raises(ValueError):
            call()
    assert st.get_current("beta") is None                     # a rejected update_current stored nothing


def test_current_state_copy_semantics():
    st = _sm(2)
    u = np.arange(8.0).reshape(4, 2) / 10
    st.set_current("u", u)                                    # default copy=True
    u[0, 0] = 99.0
    assert st.get_current("u")[0, 0] == 0.0
    got = st.get_current("u")
    got[:] = -1.0
    np.testing.assert_array_equal(st.get_current("u"), np.arange(8.0).reshape(4, 2) / 10)
    assert st.get_current("u").shape == (4, 2) and st.get_current("u").flags["C_CONTIGUOUS"]
    everything = st.get_current()
    everything["u"][:] = 7.0
    assert st.get_current("u")[1, 1] == 0.3
    # host-side keys: copy=True detaches, copy=False keeps the caller's array (state_manager.py:213-265)
    acc = np.array([0.25, 0.5])
    st.set_current("acceptance", acc, copy=True)
    acc[0] = 9.0
    assert st.get_current("acceptance")[0] == 0.25
    eff = np.array([0.1, 0.2])
    st.update_current({"efficiency": eff, "beta": 0.3}, copy=False)
    eff[1] = 0.7
    assert st.get_current("efficiency")[1] == 0.7 and st.get_current("beta") == 0.3
    st.update_current({"steps": 4, "calls": 12})
    assert st.get_current("steps") == 4 and st.get_current("calls") == 12


def test_history_views_indices_and_copies():
    st = _sm(3)
    ins = _fill(st, n_it=3, n=10)
    assert st.get_history_length() == 3
    for t, (u, x, logl) in enumerate(ins):
        np.testing.assert_array_equal(st.get_history("u", index=t), u)
        np.testing.assert_array_equal(st.get_history("x", index=t), x)
        np.testing.assert_array_equal(st.get_history("logl", index=t), logl)
        assert st.get_history("beta", index=t) == pytest.approx(0.1 * t)
    assert st.get_history("u").shape == (3, 10, 3) and st.get_history("logl").shape == (3, 10)
    assert st.get_history("u", flat=True).shape == (30, 3) and st.get_history("logl", flat=True).shape == (30,)
    np.testing.assert_array_equal(st.get_history("u", flat=True), np.concatenate([a[0] for a in ins]))
    np.testing.assert_allclose(st.get_history("beta"), [0.0, 0.1, 0.2])
    np.testing.assert_array_equal(st.get_history("iter"), [0, 1, 2])
    for bad in (3, 10, -1):
        with pytest.raises(IndexError):
            st.get_history("logl", index=bad)
    with pytest.raises(IndexError):
        st.get_history("beta", index=5)
    # what comes back is never a view of the store
    h = st.get_history("logl")
    h[:] = 0.0
    hi = st.get_history("u", index=1)
    hi[:] = 0.0
    np.testing.assert_array_equal(st.get_history("logl", index=0), ins[0][2])
    np.testing.assert_array_equal(st.get_history("u", index=1), ins[1][0])


def test_last_history_and_length_accessors():
    st = _sm(2)
    assert st.get_last_history("beta") is None and st.get_last_history("beta", default=0.0) == 0.0
    assert st.get_last_history("logl", default="none") == "none"
    ins = _fill(st, n_it=2, n=6)
    assert st.get_last_history("beta") == pytest.approx(0.1) and st.get_last_history("iter") == 1
    last = st.get_last_history("logl")
    np.testing.assert_array_equal(last, ins[1][2])
    last[:] = 0.0
    np.testing.assert_array_equal(st.get_last_history("logl"), ins[1][2])
    np.testing.assert_array_equal(st.get_last_history("u"), ins[1][0])
    assert st.get_history_length() == 2 == len(st.get_history("beta")) == len(st.get_history("logz"))
    _fill(st, n_it=1, n=6, seed=9)
    assert st.get_history_length() == 3


def test_commit_strictness():
    """state_manager.py:356-416: strict=False is the default and tolerates missing keys; strict=True wants beta and logl."""
    st = _sm(2)
    st.set_current("beta", 0.5)
    st.commit_current_to_history()                           # no logl: allowed, the scalar is recorded
    st.commit_current_to_history(strict=False)
    assert st.get_history_length() == 2 and st.get_last_history("beta") == 0.5
    st = _sm(2)
    st.set_current("logl", np.random.RandomState(0).randn(10))
    with pytest.raises(ValueError) as e:
        st.commit_current_to_history(strict=True)
    assert "beta" in str(e.value) and "required keys are missing" in str(e.value).lower()
    st = _sm(2)
    st.set_current("beta", 0.5)
    with pytest.raises(ValueError) as e:
        st.commit_current_to_history(strict=True)
    assert "logl" in str(e.value)
    st = _sm(2)
    with pytest.raises(ValueError) as e:
        st.commit_current_to_history(strict=True)
    assert "beta" in str(e.value) and "logl" in str(e.value)
    st.set_current("beta", None)
    st.set_current("logl", None)
    with pytest.raises(ValueError) as e:
        st.commit_current_to_history(strict=True)
    assert "required keys are missing" in str(e.value).lower()
    st.update_current({"beta": 0.5, "logl": np.zeros(10), "iter": 1, "calls": 100})      # optional keys ride along
    st.commit_current_to_history(strict=True)
    assert st.get_history_length() == 1 and st.get_last_history("calls") == 100


def test_logw_of_empty_single_and_multiple_iterations():
    """state_manager.py:418-480 through the interface (the numerics are pinned against the golden vectors elsewhere)."""
    st = _sm(2)
    logw, logz = st.compute_logw_and_logz(1.0)
    assert len(logw) == 0 and logz == -np.inf
    rs = np.random.RandomState(3)
    logl = rs.randn(12)
    st.update_current({"u": rs.rand(12, 2), "x": rs.rand(12, 2), "logl": logl, "beta": 0.0, "logz": 0.0, "iter": 0})
    st.commit_current_to_history()
    logw, logz = st.compute_logw_and_logz(1.0)
    from scipy.special import logsumexp
    # one iteration at beta = 0 with logZ = 0: logw_i = logl_i - 0, logZ = log mean exp(logl)
    assert logz == pytest.approx(logsumexp(logl) - np.log(12), rel=1e-12)
    assert logsumexp(logw) == pytest.approx(0.0, abs=1e-12)
    raw, _ = st.compute_logw_and_logz(1.0, normalize=False)
    np.testing.assert_allclose(raw, logl, rtol=1e-12, atol=1e-12)
    st.update_current({"u": rs.rand(12, 2), "x": rs.rand(12, 2), "logl": rs.randn(12), "beta": 0.4, "logz": -0.3, "iter": 1})
    st.commit_current_to_history()
    logw, logz = st.compute_logw_and_logz(0.8)
    assert logw.shape == (24,) and np.isfinite(logz) and logsumexp(logw) == pytest.approx(0.0, abs=1e-12)


def test_dict_round_trips(tmp_path):
    """state_manager.py:505-652: to_dict / from_dict / update_from_dict, save_state / load_state."""
    st = _sm(3)
    d0 = st.to_dict()
    assert set(d0) >= {"_current", "_history", "n_dim"} and d0["n_dim"] == 3
    st.set_current("beta", 0.5)
    d = st.to_dict()
    d["_current"]["beta"] = 999
    assert st.get_current("beta") == 0.5
    ins = _fill(st, n_it=3, n=8)
    d = st.to_dict()
    assert len(d["_history"]["beta"]) == 3 and len(d["_history"]["u"]) == 3
    np.testing.assert_array_equal(d["_history"]["logl"][2], ins[2][2])
    d["_history"]["logl"][2][:] = 0.0
    np.testing.assert_array_equal(st.get_history("logl", index=2), ins[2][2])
    from tempest_amd.state_manager import StateManager
    minimal = StateManager.from_dict({"n_dim": 5})
    assert minimal.n_dim == 5 and minimal.get_history_length() == 0
    clone = StateManager.from_dict(st.to_dict())
    assert clone.n_dim == 3 and clone.get_history_length() == 3
    np.testing.assert_array_equal(clone.get_history("u", flat=True), st.get_history("u", flat=True))
    np.testing.assert_array_equal(clone.get_current("logl"), st.get_current("logl"))
    a, za = st.compute_logw_and_logz(0.7)
    b, zb = clone.compute_logw_and_logz(0.7)
    np.testing.assert_array_equal(a, b)
    assert za == zb
    # update_from_dict merges: a partial dict touches only what it names, and drops the cached results
    other = _sm(3)
    other.set_current("calls", 7)
    res = st.compute_results()
    assert st._results_dict is res
    st.update_from_dict({"_current": {"beta": 0.9}})
    assert st._results_dict is None and st.get_current("beta") == 0.9 and st.get_history_length() == 3
    other.update_from_dict({"_current": {"beta": 0.25}})
    assert other.get_current("beta") == 0.25 and other.get_current("calls") == 7
    # file round trip; the parent directory is created
    path = tmp_path / "sub" / "run.state"
    st.save_state(path)
    assert path.exists()
    back = _sm(3)
    back.load_state(path)
    assert back.get_history_length() == 3 and back.get_current("beta") == 0.9
    np.testing.assert_array_equal(back.get_history("x", flat=True), st.get_history("x", flat=True))
    # (the loaded state forms its cached log-mixture with one streaming log-sum-exp per row, the running one folded a logaddexp
    # per iteration: the same value to rounding)
    np.testing.assert_allclose(back.compute_logw_and_logz(1.0)[0], st.compute_logw_and_logz(1.0)[0], rtol=2e-15, atol=0)


def test_results_dict():
    st = _sm(3)
    _fill(st, n_it=2, n=10)
    res = st.compute_results()
    assert isinstance(res, dict) and {"logw", "u", "x", "logl", "beta", "logz"} <= set(res)
    assert len(res["logw"]) == 20 and res["u"].shape == (2, 10, 3)
    assert st.compute_results() is res                        # cached until the state changes
    st.set_current("beta", 0.3)
    assert st._results_dict is None
