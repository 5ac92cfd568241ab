"""Generate tests/golden/*.npz by importing THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Usage (scratch cwd; the reference is read-only and never travels):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python3 /root/repo/oracle/make_golden.py

Every array written is data: seeded inputs (np.random.RandomState) and the outputs the imported
reference (minaskar/tempest v0.2.1) produced for them.  RNG-consuming reference functions are run
with np.random.{gamma,randn,choice,random,rand} wrapped so the draws they consumed are recorded
next to their outputs; the oracle / device are then checked as pure functions of those draws.
"""
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class Recorder:
    """Wrap np.random.<name> to record what the reference drew."""

    def __init__(self, *names):
        self.names = names
        self.log = {n: [] for n in names}
        self._orig = {}

    def __enter__(self):
        for n in self.names:
            self._orig[n] = getattr(np.random, n)

            def wrap(*a, _n=n, **k):
                v = self._orig[_n](*a, **k)
                self.log[_n].append(np.array(v, copy=True))
                return v
            setattr(np.random, n, wrap)
        return self

    def __exit__(self, *exc):
        for n in self.names:
            setattr(np.random, n, self._orig[n])


def history_state(rs, T, N, d, unequal=False, neg_inf=False, heavy=True):
    from tempest.state_manager import StateManager
    st = StateManager(d)
    betas = np.concatenate([[0.0], np.sort(rs.rand(T - 1)) ** 2]) if T > 1 else np.array([0.0])
    logz = np.concatenate([[0.0], -np.cumsum(rs.rand(T - 1) * 3)]) if T > 1 else np.array([0.0])
    for t in range(T):
        n = N if not unequal else max(2, N - 3 * t)
        u = rs.rand(n, d)
        x = 20 * u - 10
        logl = -0.5 * np.sum(x ** 2, axis=1) * (1.0 - 0.8 * betas[t])
        if heavy:
            logl = logl - rs.standard_cauchy(n) ** 2
        if neg_inf and t > 0:
            logl[rs.randint(n)] = -np.inf
        st.update_current({"u": u, "x": x, "logl": logl, "beta": float(betas[t]), "logz": float(logz[t]),
                           "iter": t, "calls": 0, "steps": 1, "efficiency": 1.0, "ess": 1.0, "cv": 0.0,
                           "acceptance": 1.0})
        st.commit_current_to_history()
    return st


def dump_hist(st):
    logl = st._history["logl"]
    return dict(u=st.get_history("u", flat=True), x=st.get_history("x", flat=True),
                logl=st.get_history("logl", flat=True), beta_t=np.asarray(st.get_history("beta")),
                logz_t=np.asarray(st.get_history("logz")), n_t=np.array([len(v) for v in logl]))


def g1_logw():
    out = {}
    k = 0
    for (T, N, unequal, neg_inf) in [(1, 16, False, False), (3, 16, True, False), (12, 16, False, False),
                                     (12, 64, True, True), (37, 200, False, False), (5, 1000, False, False)]:
        rs = np.random.RandomState(100 + k)
        st = history_state(rs, T, N, 3, unequal, neg_inf)
        h = dump_hist(st)
        for key in ("logl", "beta_t", "logz_t", "n_t"):
            out[f"c{k}_{key}"] = h[key]
        for j, bf in enumerate([0.0, 1e-4, 0.3, 1.0]):
            with np.errstate(all="ignore"):
                lw, lz = st.compute_logw_and_logz(bf)
                lwu, _ = st.compute_logw_and_logz(bf, normalize=False)
            out[f"c{k}_b{j}_beta"] = bf
            out[f"c{k}_b{j}_logw"] = lw
            out[f"c{k}_b{j}_logw_unnorm"] = lwu
            out[f"c{k}_b{j}_logz"] = lz
        k += 1
    out["n_cases"] = k
    np.savez_compressed(os.path.join(OUT, "g1_logw.npz"), **out)


def g2_tools():
    from tempest.tools import effective_sample_size, trim_weights, volume_variation, increment_logz
    out = {}
    rs = np.random.RandomState(7)
    for k, n in enumerate([4, 50, 1000, 20000]):
        w = np.exp(rs.randn(n) * (0.5 + k)) if k else np.ones(4)
        out[f"ess{k}_w"] = w
        out[f"ess{k}"] = effective_sample_size(w)
        for j, (ess, bins) in enumerate([(0.99, 1000), (0.9, 50), (0.999, 1000)]):
            idx, wt = trim_weights(np.arange(n), w.copy(), ess=ess, bins=bins)
            out[f"trim{k}_{j}_cfg"] = np.array([ess, bins])
            out[f"trim{k}_{j}_idx"] = idx
            out[f"trim{k}_{j}_w"] = wt
    out["inc_logz_zeros100"] = increment_logz(np.zeros(100))
    lw = rs.randn(77) * 4
    out["inc_logz_in"] = lw
    out["inc_logz_out"] = increment_logz(lw)
    # volume variation: regular, weighted, rank-deficient, too-few-samples
    x = rs.rand(500, 4)
    w = np.exp(rs.randn(500))
    out["vv0_x"], out["vv0_w"], out["vv0"] = x, w, volume_variation(x, w)
    out["vv1_x"], out["vv1"] = x, volume_variation(x)
    xr = x.copy(); xr[:, 3] = xr[:, 0] * 2.0 - xr[:, 1]
    out["vv2_x"], out["vv2_w"], out["vv2"] = xr, w, volume_variation(xr, w)
    out["vv3_x"], out["vv3"] = x[:4], volume_variation(x[:4])
    xn = 0.5 + 1e-3 * rs.randn(3000, 10)
    wn = np.exp(2 * rs.randn(3000))
    out["vv4_x"], out["vv4_w"], out["vv4"] = xn, wn, volume_variation(xn, wn)
    np.savez_compressed(os.path.join(OUT, "g2_tools.npz"), **out)


def g3_reweighter():
    from tempest.steps import Reweighter
    from tempest.state_manager import StateManager
    from tempest import config as C
    out = {}
    k = 0

    def run_case(st, n_particles, ess_ratio, vol_var, beta_prev, tag):
        nonlocal k
        st.set_current("beta", beta_prev)
        st.set_current("iter", st.get_history_length())
        rw = Reweighter(state=st, pbar=None, n_particles=n_particles, ess_ratio=ess_ratio,
                        volume_variation=vol_var, ESS_TOLERANCE=C.ESS_TOLERANCE,
                        BETA_TOLERANCE=C.BETA_TOLERANCE, BETA_RTOL=C.BETA_RTOL,
                        METRIC_ATOL=C.METRIC_ATOL, METRIC_ATOL_CV=C.METRIC_ATOL_CV)
        trace = []
        orig = rw._compute_metric_and_weights

        def traced(beta):
            trace.append(float(beta))
            return orig(beta)
        rw._compute_metric_and_weights = traced
        with np.errstate(all="ignore"):
            w = rw.run()
        h = dump_hist(st)
        for key in ("u", "logl", "beta_t", "logz_t", "n_t"):
            out[f"c{k}_{key}"] = h[key]
        out[f"c{k}_cfg"] = np.array([n_particles, ess_ratio, -1.0 if vol_var is None else vol_var, beta_prev])
        out[f"c{k}_tag"] = tag
        out[f"c{k}_trace"] = np.array(trace)
        out[f"c{k}_weights"] = w
        out[f"c{k}_out"] = np.array([st.get_current("beta"), st.get_current("ess"),
                                     st.get_current("logz"), st.get_current("cv")], dtype=float)
        k += 1

    # (i) tests/test_steps.py:100-145 — ESS == target at beta=0 -> stay
    np.random.seed(42)
    st = StateManager(2)
    for key, v in (("iter", 0), ("beta", 0.0), ("logz", 0.0), ("calls", 0)):
        st.set_current(key, v)
    u = np.random.rand(16, 2); x = u * 6 - 3
    st.update_current({"u": u, "x": x, "logl": -np.sum(x ** 2, axis=1), "beta": 0.0, "logz": 0.0})
    st.commit_current_to_history()
    run_case(st, 16, 1.0, None, 0.0, "ref_test_stay_at_zero")
    # (ii) tests/test_steps.py:147-200 — bisection reaches |ESS-16| <= 0.5
    np.random.seed(123)
    st = StateManager(10)
    for key, v in (("iter", 0), ("beta", 0.0), ("logz", 0.0), ("calls", 0)):
        st.set_current(key, v)
    for _ in range(3):
        u = np.random.rand(32, 10); x = u * 20 - 10
        logl = -0.5 * np.sum(x ** 2, axis=1) - 0.5 * 10 * np.log(2 * np.pi)
        st.update_current({"u": u, "x": x, "logl": logl, "beta": 0.0, "logz": 0.0})
        st.commit_current_to_history()
    run_case(st, 32, 0.5, None, 0.0, "ref_test_bisection_converges")
    # (iii) synthetic histories, ESS and dynamic modes, from several beta_prev
    for seed, T, N, d, er, vv in [(1, 4, 64, 3, 2.0, None), (2, 9, 64, 3, 2.0, None),
                                  (3, 15, 100, 5, 1.0, None), (4, 6, 64, 3, 2.0, 0.05),
                                  (5, 9, 64, 3, 2.0, 0.2), (6, 12, 80, 4, 1.5, 1.0),
                                  (7, 20, 50, 2, 4.0, None)]:
        rs = np.random.RandomState(seed)
        st = history_state(rs, T, N, d, heavy=False)
        bp = float(st.get_history("beta")[-1])
        run_case(st, N, er, vv, bp, f"synthetic_seed{seed}")
    out["n_cases"] = k
    np.savez_compressed(os.path.join(OUT, "g3_reweighter.npz"), **out)


def g4_resample():
    from tempest.tools import systematic_resample
    out = {}
    rs = np.random.RandomState(11)
    k = 0
    for n, size in [(4, 4), (100, 37), (5000, 1000), (1000, 4000)]:
        w = np.exp(rs.randn(n) * 2)
        w /= w.sum()
        if k == 0:
            w = np.array([0.6, 0.2, 0.15, 0.05])
        for seed in (0, 1):
            np.random.seed(seed)
            u0 = np.random.random()
            np.random.seed(seed)
            idx = systematic_resample(size, w)
            out[f"c{k}_w"], out[f"c{k}_size"], out[f"c{k}_u0"], out[f"c{k}_idx"] = w, size, u0, idx
            k += 1
    # unnormalised weights branch (tools.py:214-215)
    w = np.exp(rs.randn(300)); np.random.seed(5); u0 = np.random.random(); np.random.seed(5)
    out[f"c{k}_w"], out[f"c{k}_size"], out[f"c{k}_u0"], out[f"c{k}_idx"] = w, 128, u0, systematic_resample(128, w)
    k += 1
    out["n_cases"] = k
    # multinomial: np.random.choice(p=w) with the uniforms it consumed
    w = np.exp(rs.randn(2000) * 1.5); w /= w.sum()
    np.random.seed(3); uu = np.random.random_sample(500); np.random.seed(3)
    out["mult_w"], out["mult_u"] = w, uu
    out["mult_idx"] = np.random.choice(np.arange(2000), size=500, replace=True, p=w)
    np.savez_compressed(os.path.join(OUT, "g4_resample.npz"), **out)


def g5_boundaries():
    from tempest.mcmc import apply_boundary_conditions, check_bounds
    out = {}
    rs = np.random.RandomState(5)
    u = rs.randn(400, 6) * 1.5 + 0.5
    per, ref = np.array([0, 3]), np.array([1, 4])
    out["u"], out["periodic"], out["reflective"] = u, per, ref
    out["applied"] = apply_boundary_conditions(u, per, ref)
    out["ok_after"] = check_bounds(out["applied"], per, ref)
    out["ok_raw_nobc"] = check_bounds(u)
    out["ok_raw_bc"] = check_bounds(u, per, ref)
    np.savez_compressed(os.path.join(OUT, "g5_boundaries.npz"), **out)


def g6_mcmc():
    from tempest.mcmc import TPCNRunner, RWMRunner
    from tempest.modes import ModeStatistics
    out = {}
    rs = np.random.RandomState(21)
    d, n, K = 5, 96, 3
    means = 0.5 + 0.05 * rs.randn(K, d)
    covs = np.empty((K, d, d))
    for c in range(K):
        A = rs.randn(d, d) * 0.02
        covs[c] = A @ A.T + 1e-4 * np.eye(d)
    dof = np.array([1e6, 5.0, 30.0])
    ms = ModeStatistics(means, covs, dof)
    u = 0.5 + 0.03 * rs.randn(n, d)
    x = 20 * u - 10
    logl = -0.5 * np.sum(x ** 2, axis=1)
    assign = rs.randint(K, size=n)
    ll = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    pt = lambda uu: 20 * uu - 10
    for name, cls in (("tpcn", TPCNRunner), ("rwm", RWMRunner)):
        r = cls(u, x, logl, None, assign, 0.7, ms, ll, pt, None, 2, 40, None, None, False)
        r.sigmas = r.sigmas * np.array([1.0, 0.6, 0.3])
        out[f"{name}_sigmas"] = r.sigmas.copy()
        props, gam, zz = [], [], []
        for kk in range(n):
            with Recorder("gamma", "randn") as rec:
                p = r._propose(kk)
            assert len(rec.log["randn"]) == 1, "fixture must be single-attempt"
            props.append(p); zz.append(rec.log["randn"][0])
            if name == "tpcn":
                gam.append(rec.log["gamma"][0])
        out[f"{name}_proposal"] = np.array(props)
        out[f"{name}_z"] = np.array(zz)
        if name == "tpcn":
            out["tpcn_gamma"] = np.array(gam, dtype=float)
        up = np.array(props)
        lp = -0.5 * np.sum((20 * up - 10) ** 2, axis=1)
        out[f"{name}_factor"] = r._compute_acceptance_factor(up, lp)
        # sigma adaptation + adaptive steps tables
        tab = []
        for it in (1, 2, 7, 50):
            for ma in (0.0, 0.1, 0.234, 0.6, 1.0):
                r2 = cls(u, x, logl, None, assign, 0.7, ms, ll, pt, None, 2, 40, None, None, False)
                r2.iteration = it
                r2.sigmas[:] = out[f"{name}_sigmas"]
                r2._adapt_sigma(1, ma)
                tab.append([it, ma, r2.sigmas[1]])
        out[f"{name}_adapt"] = np.array(tab)
        tab = []
        for acc in (0.0, 0.005, 0.1, 0.234, 0.9):
            for scale in (1e-8, 0.1, 1.0, 3.0):
                r2 = cls(u, x, logl, None, assign, 0.7, ms, ll, pt, None, 2, 40, None, None, False)
                r2.sigmas = out[f"{name}_sigmas"] * scale
                tab.append([acc, scale, r2._calculate_adaptive_steps(acc)])
        out[f"{name}_steps"] = np.array(tab)
    for key, v in (("u", u), ("logl", logl), ("assign", assign), ("means", means), ("covs", ms.covariances),
                   ("chol", ms.chol_covariances), ("inv", ms.inv_covariances), ("dof", dof)):
        out[key] = v
    out["beta"] = 0.7
    out["n_steps_n_max"] = np.array([2, 40])
    np.savez_compressed(os.path.join(OUT, "g6_mcmc.npz"), **out)


def g7_modes():
    from tempest.student import fit_mvstud
    from tempest.modes import ModeStatistics
    out = {}
    rs = np.random.RandomState(31)
    k = 0
    for n, d, kind in [(400, 3, "gauss"), (300, 4, "t3"), (2000, 10, "narrow"), (50, 6, "degenerate")]:
        if kind == "gauss":
            data = rs.randn(n, d) * 0.1 + 0.5
        elif kind == "t3":
            data = rs.standard_t(3, size=(n, d)) * 0.05 + 0.4
        elif kind == "narrow":
            data = 0.5 + 1e-3 * rs.randn(n, d) @ np.triu(rs.rand(d, d))
        else:
            data = np.tile(rs.rand(1, d), (n, 1))
            data[:, 0] += 1e-3 * rs.randn(n)
        mu, Sig, nu = fit_mvstud(data)
        out[f"fit{k}_data"], out[f"fit{k}_mu"], out[f"fit{k}_Sigma"], out[f"fit{k}_nu"] = data, mu, Sig, nu
        k += 1
    out["n_fit"] = k
    # from_global with the up-sampling indices it drew
    u = rs.rand(600, 5) * 0.2 + 0.4
    w = np.exp(rs.randn(600) * 1.5)
    with Recorder("choice") as rec:
        ms = ModeStatistics.from_global(u, w)
    out["fg_u"], out["fg_w"], out["fg_idx"] = u, w, rec.log["choice"][0]
    out["fg_means"], out["fg_covs"], out["fg_chol"], out["fg_inv"], out["fg_dof"] = (
        ms.means, ms.covariances, ms.chol_covariances, ms.inv_covariances, ms.degrees_of_freedom)
    # from_particles, 3 labels
    labels = rs.randint(3, size=600)
    with Recorder("choice") as rec:
        ms = ModeStatistics.from_particles(u, w, labels)
    out["fp_labels"] = labels
    for c in range(3):
        out[f"fp_idx{c}"] = rec.log["choice"][c]
    out["fp_means"], out["fp_covs"], out["fp_chol"], out["fp_inv"], out["fp_dof"] = (
        ms.means, ms.covariances, ms.chol_covariances, ms.inv_covariances, ms.degrees_of_freedom)
    np.savez_compressed(os.path.join(OUT, "g7_modes.npz"), **out)


def g8_inf_repair():
    from tempest.steps import Mutator
    from tempest.state_manager import StateManager
    from tempest.modes import ModeStatistics
    out = {}
    n, d = 64, 3
    st = StateManager(d)
    for key, v in (("iter", 1), ("beta", 0.0), ("logz", 0.0), ("calls", 0)):
        st.set_current(key, v)

    def ll(x):
        l = -0.5 * np.sum(x ** 2, axis=1)
        l[x[:, 0] > 5.0] = -np.inf
        l[x[:, 1] < -8.0] = np.inf
        return l, None
    mut = Mutator(st, lambda u: 20 * u - 10, ll, None, n_particles=n, n_dim=d)
    np.random.seed(9)
    with Recorder("rand", "choice") as rec:
        mut.run(ModeStatistics(np.zeros((1, d)), np.eye(d)[None], np.array([1e6])))
    out["u_drawn"] = rec.log["rand"][0]
    out["choice"] = rec.log["choice"][0]
    out["u_final"], out["x_final"], out["logl_final"] = (st.get_current("u"), st.get_current("x"),
                                                         st.get_current("logl"))
    out["logz"] = st.get_current("logz")
    out["calls"] = st.get_current("calls")
    np.savez_compressed(os.path.join(OUT, "g8_inf_repair.npz"), **out)


def g10_cluster():
    """HierarchicalGaussianMixture / GaussianMixture of the reference on seeded synthetic sets: number of clusters,
    labels, fitted two-component parameters, BICs (distributional pins: the seeding RNG cannot be reproduced)."""
    from tempest.cluster import GaussianMixture, HierarchicalGaussianMixture
    out = {}
    rs = np.random.RandomState(77)
    sets = {}
    # two well separated blobs in 3-D, weighted
    a = 0.3 + 0.03 * rs.randn(300, 3); b = 0.7 + 0.04 * rs.randn(200, 3)
    sets["two"] = (np.clip(np.vstack([a, b]), 0, 1), np.exp(0.3 * rs.randn(500)), np.r_[np.zeros(300), np.ones(200)])
    # four modes in 4-D (config-3 shape: +-offset in the first two coordinates)
    pts, lab = [], []
    for k, (sx, sy) in enumerate([(-1, -1), (-1, 1), (1, -1), (1, 1)]):
        c = np.array([0.5 + 0.2 * sx, 0.5 + 0.2 * sy, 0.5, 0.5])
        pts.append(c + 0.02 * rs.randn(250, 4)); lab.append(np.full(250, k))
    sets["four"] = (np.vstack(pts), np.ones(1000), np.concatenate(lab))
    # one Gaussian blob: must not split
    sets["one"] = (0.5 + 0.05 * rs.randn(600, 3), np.exp(0.5 * rs.randn(600)), np.zeros(600))
    for name, (X, w, truth) in sets.items():
        out[f"{name}_X"], out[f"{name}_w"], out[f"{name}_truth"] = X, w, truth
        h = HierarchicalGaussianMixture(n_init=1, max_iterations=1000, min_points=None, threshold_modifier=1.0,
                                        covariance_type="full", verbose=False, normalize=True)
        h.fit(X, w)
        out[f"{name}_K"] = h.n_clusters_
        out[f"{name}_labels"] = h.labels_
        out[f"{name}_pred"] = h.predict(X)
        out[f"{name}_weights"] = h.cluster_weights_
        out[f"{name}_centers"] = np.array(h.cluster_centers_)
    X, w, _ = sets["two"]
    for K in (1, 2):
        g = GaussianMixture(n_components=K, random_state=42).fit(X, w)
        order = np.argsort(g.means_[:, 0])
        out[f"gmm{K}_weights"], out[f"gmm{K}_means"], out[f"gmm{K}_covs"] = g.weights_[order], g.means_[order], g.covariances_[order]
        out[f"gmm{K}_lower"], out[f"gmm{K}_bic"], out[f"gmm{K}_niter"] = g.lower_bound_, g.bic(X), g.n_iter_
        out[f"gmm{K}_pred"] = np.argsort(order)[g.predict(X)]
    np.savez_compressed(os.path.join(OUT, "g10_cluster.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    import tempest
    print("reference", tempest.__version__, "numpy", np.__version__)
    for fn in (g1_logw, g2_tools, g3_reweighter, g4_resample, g5_boundaries, g6_mcmc, g7_modes, g8_inf_repair, g10_cluster):
        fn()
        print("wrote", fn.__name__)


if __name__ == "__main__":
    sys.exit(main())
