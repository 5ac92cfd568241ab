"""BASELINE.json configurations 2 and 3 end to end on one MI355X (SURVEY 8d targets with analytic evidence)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def prior20(u):
    return 20 * u - 10


def c2_target(dev, d=50):
    """50-D zero-mean correlated Gaussian, Sigma = A A^T / d + 0.5 I, A ~ N(0,1) from RandomState(1), normalised."""
    A = np.random.RandomState(1).randn(d, d)
    S = A @ A.T / d + 0.5 * np.eye(d)
    P = torch.from_numpy(np.linalg.inv(S)).to(dev)
    const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))

    def loglike(x):
        return -0.5 * ((x @ P) * x).sum(dim=1) + const
    return loglike, S


def test_config2_gauss50_65536_rwm():
    """50-D correlated Gaussian, 65 536 particles, random-walk mutation: analytic logZ = -50 ln 20 = -149.79.
    At its default settings (ess_ratio 2, n_steps 1) the ALGORITHM overestimates the evidence in 50-D: the NumPy
    oracle sampler (the reference's algorithm restated) gives logZ = -144.482 at N=512 with seed 0 and the device run
    with the same seed gives -144.482 as well; the excess shrinks with N (about +2.5 here).  The gate is therefore the
    posterior moments plus an evidence window that contains that known bias."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    loglike, S = c2_target(dev)
    t0 = time.time()
    s = tp.Sampler(prior20, loglike, 50, n_particles=65536, vectorize=True, clustering=False, sample="rwm",
                   random_state=0, backend="torch", batch_prior=True)
    s.run(n_total=4 * 65536, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    x, w, _ = s.posterior()
    mean = np.average(x, weights=w, axis=0)
    var = np.average((x - mean) ** 2, weights=w, axis=0)
    steps = np.asarray(s.state.get_history("steps")); beta = np.asarray(s.state.get_history("beta"))
    pms = steps[beta > 0].sum() * 65536
    print(f"config2: logZ={logz:.3f} (analytic {-50 * np.log(20):.3f}) iters={len(beta)} wall={wall:.1f}s pms/s={pms / wall:.3g}")
    # the algorithm's own excess at this size: +2.45 (seed 0; +2.5 at the 16 384-particle twin, +5.3 at N = 512 in device and
    # oracle alike); window = that value +- a few seed-sigma of a 65 536-particle run
    assert 1.5 < logz + 50 * np.log(20.0) < 3.3
    np.testing.assert_allclose(mean, 0.0, atol=0.1)
    np.testing.assert_allclose(var, np.diag(S), rtol=0.1)
    assert pms / wall > 5e7           # 7.6e7 in round 2, 9.2e7 with the row-walker kernel (first process on a fresh box: lower)


def test_config2_gauss50_65536_tpcn():
    """Config 2 with the default kernel (t-preconditioned Crank-Nicolson): never more than ~17 redraw attempts per particle,
    46 iterations as with RWM; evidence excess +1.56 at this size."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    loglike, S = c2_target(dev)
    t0 = time.time()
    s = tp.Sampler(prior20, loglike, 50, n_particles=65536, vectorize=True, clustering=False, sample="tpcn",
                   random_state=0, backend="torch", batch_prior=True)
    s.run(n_total=4 * 65536, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    x, w, _ = s.posterior()
    mean = np.average(x, weights=w, axis=0)
    var = np.average((x - mean) ** 2, weights=w, axis=0)
    steps = np.asarray(s.state.get_history("steps")); beta = np.asarray(s.state.get_history("beta"))
    pms = steps[beta > 0].sum() * 65536
    print(f"config2 tpCN: logZ={logz:.3f} (analytic {-50 * np.log(20):.3f}) iters={len(beta)} wall={wall:.1f}s pms/s={pms / wall:.3g}")
    assert beta[-1] == 1.0 and 40 <= len(beta) <= 52
    assert 0.8 < logz + 50 * np.log(20.0) < 2.4
    np.testing.assert_allclose(mean, 0.0, atol=0.1)
    np.testing.assert_allclose(var, np.diag(S), rtol=0.1)
    assert pms / wall > 7e7


def test_config3_mixture32_clustering():
    """32-D four-mode Gaussian mixture (modes at (+-4, +-4, 0, ...), sigma 0.5), clustering=True; analytic
    logZ = -32 ln 20.  65 536 particles here (BASELINE's 262 144 is run by scripts, not in the test suite)."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d = 32
    mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
    for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
        mus[k, 0], mus[k, 1] = a, b
    const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

    def loglike(x):
        q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)          # (n, 4)
        return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
    t0 = time.time()
    s = tp.Sampler(prior20, loglike, d, n_particles=65536, vectorize=True, clustering=True, random_state=0,
                   backend="torch", batch_prior=True)
    s.run(n_total=4 * 65536, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    x, w, _ = s.posterior()
    occ = [float(np.sum(w[(np.sign(x[:, 0]) == a) & (np.sign(x[:, 1]) == b)])) for a in (-1, 1) for b in (-1, 1)]
    print(f"config3: logZ={logz:.3f} (analytic {-d * np.log(20):.3f}) iters={len(s.state.get_history('beta'))} "
          f"wall={wall:.1f}s K={s._core.trainer.clusterer.n_clusters_} mode occupancy={np.round(occ, 3)}")
    # the algorithm's evidence excess at default settings grows with dimension (+0.2 at d=10, ~+1 at d=32, see config 2):
    # +0.985 at this size and seed, +0.99 at 262 144, +1.2 at the N = 1024 twin (reference ensemble -94.68 +- 0.12)
    assert 0.4 < logz + d * np.log(20.0) < 1.7
    assert min(occ) > 0.15 and max(occ) < 0.35


def test_config5_funnel100_small():
    """100-D Neal funnel (v = x0 ~ N(0, 3^2), x_i | v ~ N(0, e^v)), per-dimension affine prior, tpCN;
    analytic logZ = -ln 30 - 99 ln 600 = -636.70.  4 096 particles here (BASELINE: 2 097 152 over 8 GPUs): exercises the
    d = 100 kernels (multi-lane proposal with one staged matrix slot, LDS-tiled covariance, 100 x 100 Cholesky/inverse).
    One global Gaussian-preconditioned mode does not resolve the funnel's neck -- in the REFERENCE either: its own
    run of this twin (3 seeds, 3.4 CPU-hours each, tests/golden/ref_ensembles.json) gives logZ = -639.44 +- 0.09 in 31
    iterations with posterior mean of v = 8.55.  The GPU run must land on the reference, not on the analytic value."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d = 100
    scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
    shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

    def prior(u):
        return u * scale + shift

    def loglike(x):
        v = x[:, 0]
        lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
        lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
        return lv + lr
    t0 = time.time()
    s = tp.Sampler(prior, loglike, d, n_particles=4096, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True)
    s.run(n_total=4 * 4096, progress=False)
    logz = s.evidence()[0]
    truth = -np.log(30.0) - 99 * np.log(600.0)
    print(f"config5 (N=4096): logZ={logz:.3f} (analytic {truth:.3f}) iters={len(s.state.get_history('beta'))} wall={time.time() - t0:.1f}s")
    assert np.isfinite(logz) and abs(logz - truth) < 8.0
    import json
    import os
    ref = [r for r in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_ensembles.json")))["runs"]
           if r["config"] == "c5twin_funnel100_n4096"]
    assert len(ref) >= 3
    ref_logz = np.array([r["logz"] for r in ref])
    assert abs(logz - ref_logz.mean()) < 3 * max(ref_logz.std(ddof=1), 0.1) + 0.1, (logz, ref_logz)
    assert abs(len(s.state.get_history("beta")) - np.mean([r["iters"] for r in ref])) <= 2
    x, w, _ = s.posterior()
    assert abs(np.average(x[:, 0], weights=w) - np.mean([r["mean"][0] for r in ref])) < 0.5
    # x_i | v has standard deviation e^{v/2} (up to ~90 in the mouth): gate the means in units of the marginal scale
    print("config5 posterior mean of v:", np.average(x[:, 0], weights=w))
    assert np.all(np.abs(np.average(x[:, 1:], weights=w, axis=0)) < 30.0)


def _ref(name):
    import json
    import os
    runs = [r for r in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_ensembles.json")))["runs"]
            if r["config"] == name]
    lz = np.array([r["logz"] for r in runs])
    return lz.mean(), lz.std(ddof=1), runs


def test_config2_twin_matches_reference_ensemble():
    """BASELINE config 2's target (50-D correlated Gaussian) at N = 512 with the tpCN kernel, the size the reference can run
    (its RWM variant spends hours in the per-walker redraw loop): reference ensemble -147.95 +- 0.15 in 46 iterations (analytic
    -149.79: the algorithm's own evidence excess at d = 50); 6 GPU seeds must land on it."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    mu, sd, runs = _ref("c2twin_gauss50_n512_tpcn")
    d = 50
    A = np.random.RandomState(1).randn(d, d)
    S = A @ A.T / d + 0.5 * np.eye(d)
    P = torch.from_numpy(np.linalg.inv(S)).to(dev)
    const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
    got, its = [], []
    for seed in range(6):
        s = tp.Sampler(prior20, lambda x: -0.5 * ((x @ P) * x).sum(dim=1) + const, d, vectorize=True, n_particles=512,
                       clustering=False, random_state=seed)
        s.run(n_total=2048, progress=False)
        got.append(s.evidence()[0]); its.append(len(s.state.get_history("beta")))
    got = np.array(got)
    print("config2 twin: ref", mu, sd, "gpu", got.mean(), got.std(ddof=1), "iters", its)
    assert np.all(np.abs(got - mu) <= 3 * sd + 0.15), (got, mu, sd)
    assert abs(got.mean() - mu) < 3 * sd / np.sqrt(6) + 0.1
    assert abs(np.mean(its) - np.mean([r["iters"] for r in runs])) <= 2


def test_config3_twin_matches_reference_ensemble():
    """BASELINE config 3's target (32-D four-mode mixture, clustering=True) at N = 1024: reference ensemble -94.68 +- 0.12 in
    50 iterations (analytic -95.86)."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    mu, sd, runs = _ref("c3twin_mix32_n1024_cluster")
    d = 32
    mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
    for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
        mus[k, 0], mus[k, 1] = a, b
    const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

    def loglike(x):
        q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
        return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
    got, its = [], []
    for seed in range(6):
        s = tp.Sampler(prior20, loglike, d, vectorize=True, n_particles=1024, clustering=True, random_state=seed)
        s.run(n_total=4096, progress=False)
        got.append(s.evidence()[0]); its.append(len(s.state.get_history("beta")))
    got = np.array(got)
    print("config3 twin: ref", mu, sd, "gpu", got.mean(), got.std(ddof=1), "iters", its)
    assert np.all(np.abs(got - mu) <= 3 * sd + 0.15), (got, mu, sd)
    assert abs(got.mean() - mu) < 3 * sd / np.sqrt(6) + 0.1
    assert abs(np.mean(its) - np.mean([r["iters"] for r in runs])) <= 3


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json's configurations at THEIR sizes on one MI355X: the sizes the kernels are tuned for.
def test_config3_fullsize_262144_clustering():
    """Config 3 as stated: 32-D four-mode mixture, 262 144 particles, clustering=True.  Evidence window as the 65 536 run
    (the algorithm's own excess at d = 32 is about +1), the four modes equally occupied, cluster count as the reference's
    twin ensemble (K = 1: the BIC on the unweighted likelihood does not split this target)."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d, n = 32, 262144
    mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
    for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
        mus[k, 0], mus[k, 1] = a, b
    const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

    def loglike(x):
        q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
        return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
    t0 = time.time()
    s = tp.Sampler(prior20, loglike, d, n_particles=n, vectorize=True, clustering=True, random_state=0,
                   backend="torch", batch_prior=True)
    s.run(n_total=4 * n, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    beta = np.asarray(s.state.get_history("beta")); steps = np.asarray(s.state.get_history("steps"))
    x, w, _ = s.posterior()
    occ = [float(np.sum(w[(np.sign(x[:, 0]) == a) & (np.sign(x[:, 1]) == b)])) for a in (-1, 1) for b in (-1, 1)]
    print(f"config3 full size: logZ={logz:.3f} (analytic {-d * np.log(20):.3f}) iters={len(beta)} wall={wall:.1f}s "
          f"pms/s={steps[beta > 0].sum() * n / wall:.3g} K={s._core.trainer.clusterer.n_clusters_} occupancy={np.round(occ, 3)}")
    assert wall < 60.0
    assert np.all(np.diff(beta) >= 0) and beta[-1] == 1.0
    # K: the reference's split search does not split large clean multi-mode sets (tests/golden/ref_hgm_scale.json: K = 1 at
    # 100 000 rows, improvement -6 315 against a threshold of 6 459; its twin at N = 1024 ends with K = 1 or 2:
    # ref_cluster_counts.json); the device search makes the same decisions (test_split_decisions_on_growing_sets_...)
    assert s._core.trainer.clusterer.n_clusters_ in (1, 2)
    assert 0.4 < logz + d * np.log(20.0) < 1.6            # +0.99 at this size (see the 65 536-particle test above)
    assert min(occ) > 0.22 and max(occ) < 0.28
    np.testing.assert_allclose(np.average(x[:, 2:], weights=w, axis=0), 0.0, atol=0.05)
    np.testing.assert_allclose(np.average(x[:, 2:] ** 2, weights=w, axis=0), 0.25, rtol=0.1)


def test_four_modes_at_config_scale_run_with_k_gt_1_on_the_matrix_core_rounds(monkeypatch):
    """A BASELINE-scale run with SEVERAL proposal modes above 16 dimensions (the review's gap: config 3 itself ends with K = 1,
    like the reference): the separable 32-D four-mode target of tests/golden/ref_cluster_counts.json at 262 144 particles, the
    clustering working set thinned to 4 096 rows so that the split search finds the modes (at full size it does not -- there
    as in the reference, ref_hgm_scale.json).  K grows to 3-4, the steps of a few attempts per particle run as matrix-core
    rounds over mode-pure tiles (propose_blkm.hip, several modes), and the answer is the one-mode run's: evidence inside the
    same window, the four modes equally occupied."""
    import tempest_amd as tp
    from tempest_amd import mcmc
    dev = torch.device("cuda", 0)
    d, n = 32, 262144
    mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
    for k, (a, b) in enumerate([(-6, -6), (-6, 6), (6, -6), (6, 6)]):
        mus[k, 0], mus[k, 1] = a, b
    const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.09))

    def loglike(x):
        q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
        return torch.logsumexp(-0.5 * q / 0.09, dim=1) + const
    seen = []                                      # (K, rounds of the blocked path) whenever the regime rule runs
    orig = mcmc.StepEngine._regime

    def spy(self, mean_attempts):
        orig(self, mean_attempts)
        seen.append((self.K, self.blocked))
    monkeypatch.setattr(mcmc.StepEngine, "_regime", spy)
    from tempest_amd.steps import train as tr
    ks = []
    orig_run = tr.Trainer.run

    def trun(self, w):
        ms = orig_run(self, w)
        ks.append(int(ms.K))
        return ms
    monkeypatch.setattr(tr.Trainer, "run", trun)
    t0 = time.time()
    s = tp.Sampler(prior20, loglike, d, n_particles=n, vectorize=True, clustering=True, random_state=0,
                   backend="torch", batch_prior=True)
    s._core.trainer.clusterer.max_points = 4096
    s.run(n_total=4 * n, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    x, w, _ = s.posterior()
    occ = [float(np.sum(w[(np.sign(x[:, 0]) == a) & (np.sign(x[:, 1]) == b)])) for a in (-1, 1) for b in (-1, 1)]
    print(f"four modes, K > 1: logZ={logz:.3f} (analytic {-d * np.log(20):.3f}) iters={len(ks)} wall={wall:.1f}s K max {max(ks)} "
          f"last {ks[-5:]} occupancy={np.round(occ, 3)}")
    assert max(ks) >= 3
    assert any(K >= 2 and blocked >= 1 for K, blocked in seen)      # several modes DID run on the blocked rounds
    assert 0.4 < logz + d * np.log(20.0) < 1.6
    assert min(occ) > 0.22 and max(occ) < 0.28
    assert wall < 90.0


def test_config4_fullsize_1048576_rosenbrock():
    """Config 4's ensemble on ONE GPU (it fits): 10-D README Rosenbrock, 1 048 576 particles, clustering=False.  logZ within
    3 sigma_ref of the reference's `c1_rosenbrock_nocluster` ensemble (16 seeds at N = 1000: -29.804 +- 0.115; analytic
    -29.990), posterior moments against the closed form (E x_even = 1, Var 0.5; E x_odd = 1.5, Var 2.55) at the accuracy the
    reference ensemble itself reaches."""
    import tempest_amd as tp
    mu, sd, runs = _ref("c1_rosenbrock_nocluster")
    n = 1048576

    def like(x):
        return -(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0).sum(dim=1)
    t0 = time.time()
    s = tp.Sampler(prior20, like, 10, n_particles=n, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True)
    s.run(n_total=4 * n, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    beta = np.asarray(s.state.get_history("beta")); steps = np.asarray(s.state.get_history("steps"))
    x, w, _ = s.posterior()
    mean = np.average(x, weights=w, axis=0)
    var = np.average((x - mean) ** 2, weights=w, axis=0)
    print(f"config4 full size: logZ={logz:.4f} (reference ensemble {mu:.3f} +- {sd:.3f}, analytic -29.990) iters={len(beta)} "
          f"wall={wall:.1f}s pms/s={steps[beta > 0].sum() * n / wall:.3g} mean={np.round(mean, 3)} var={np.round(var, 3)}")
    assert wall < 60.0
    assert abs(logz - mu) <= 3 * sd
    assert abs(len(beta) - np.mean([r["iters"] for r in runs])) <= 3
    np.testing.assert_allclose(mean[::2], 1.0, atol=0.05)
    np.testing.assert_allclose(mean[1::2], 1.5, atol=0.08)
    np.testing.assert_allclose(var[::2], 0.5, rtol=0.08)
    np.testing.assert_allclose(var[1::2], 2.55, rtol=0.10)


def test_config5_shard_262144_funnel100_first_iterations():
    """Config 5's single-GPU shard (2 097 152 / 8 = 262 144 particles, 100-D funnel, tpCN) for a fixed number of iterations:
    the d = 100 kernels at the size they are tuned for.  The whole run is dominated by the reference's redraw-until-in-bounds
    rule (hundreds of attempts per particle and step near the prior), so the gate is on the first annealing iterations:
    finite evidence, a monotone schedule that has left beta = 0, sane acceptance, and the redraw probe of the proposal kernel
    (mean attempts per particle of its first block) recorded."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d, n = 100, 262144
    scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
    shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

    def prior(u):
        return u * scale + shift

    def loglike(x):
        v = x[:, 0]
        lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
        lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
        return lv + lr
    s = tp.Sampler(prior, loglike, d, n_particles=n, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True)
    t0 = time.time()
    attempts = []
    while True:
        s.sample(return_state=False)
        beta = s.state.get_current("beta")
        if beta > 0.0:
            eng = next(iter(s._core.mutator._engines.values()), None)
            if eng is not None:
                attempts.append(float(eng.mailbox_np[:, 6].max()))
        if beta > 0.0 and len(attempts) >= 2:
            break
        assert time.time() - t0 < 120.0
    wall = time.time() - t0
    st = s.state
    betas = np.asarray(st.get_history("beta")); steps = np.asarray(st.get_history("steps")); acc = np.asarray(st.get_history("acceptance"))
    logz = np.asarray(st.get_history("logz"))
    print(f"config5 shard: iterations={len(betas)} beta={betas[-2:]} steps={steps[-2:]} acceptance={np.round(acc[-2:], 3)} "
          f"logz={logz[-1]:.3f} attempts/particle (probe, max over the run's steps)={attempts} wall={wall:.1f}s "
          f"pms/s={steps[betas > 0].sum() * n / wall:.3g}")
    assert wall < 60.0
    assert np.all(np.isfinite(logz)) and np.all(np.diff(betas) >= 0) and 0.0 < betas[-1] < 1.0
    assert np.all(steps[betas > 0] >= d)                      # the adaptive rule's floor n_steps * n_dim (mcmc.py:119-131)
    assert np.all((acc[betas > 0] > 0.02) & (acc[betas > 0] < 0.9))
    assert all(a >= 1.0 for a in attempts)
    assert st.ctx.size == n * len(betas)
    # the ensemble is still inside the prior box and the likelihoods are finite
    u = st.dev("u")
    assert bool(((u >= 0) & (u <= 1)).all()) and bool(torch.isfinite(st.dev("logl")).all())


def test_config5_shard_131072_funnel100_to_beta_one():
    """A config-5 shard run to the end: 131 072 particles of the 100-D funnel (tpCN) until beta = 1 and the stopping rule.
    The reference's own run of the N = 4096 twin (3 seeds, 3.4 CPU-hours each) takes 31 iterations and ends with a posterior
    mean of v = 8.55 (logZ -639.44 +- 0.09); this build's twin 31 iterations, 8.51.  A shard 32 times larger must walk the
    same schedule -- the annealing schedule is set by ESS ratios, which do not depend on N -- and land on the same posterior."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d, n = 100, 131072
    scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
    shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

    def prior(u):
        return u * scale + shift

    def loglike(x):
        v = x[:, 0]
        lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
        lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
        return lv + lr
    s = tp.Sampler(prior, loglike, d, n_particles=n, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True)
    t0 = time.time()
    s.run(n_total=4 * n, progress=False)
    wall = time.time() - t0
    logz = s.evidence()[0]
    beta = np.asarray(s.state.get_history("beta")); steps = np.asarray(s.state.get_history("steps"))
    x, w, _ = s.posterior()
    mean_v = float(np.average(x[:, 0], weights=w))
    mu, sd, runs = _ref("c5twin_funnel100_n4096")
    ref_it = np.mean([r["iters"] for r in runs])
    print(f"config5 shard to beta=1: logZ={logz:.3f} (twin reference {mu:.2f} +- {sd:.2f}, analytic -636.70) iters={len(beta)} "
          f"(reference twin {ref_it:.0f}) steps={int(steps[beta > 0].sum())} mean v={mean_v:.3f} wall={wall:.1f}s "
          f"pms/s={steps[beta > 0].sum() * n / wall:.3g}")
    assert wall < 60.0 and beta[-1] == 1.0 and np.all(np.diff(beta) >= 0)
    assert abs(len(beta) - ref_it) <= 3
    assert abs(mean_v - np.mean([r["mean"][0] for r in runs])) < 0.4      # reference twin: 8.55
    assert abs(logz - mu) < 1.5             # same estimator, larger N: the twin's value up to the N-dependence of its bias


def test_config5_full_2097152_funnel100_on_one_gpu():
    """BASELINE config 5 at its STATED ensemble on one MI355X: 2 097 152 particles of the 100-D funnel (tpCN) to beta = 1 --
    3.4 GB of history per iteration, ~100 GB at the end (tempest/state_manager.py:356-416 appends without bound).
    u and x live in a mapped address range that grows in place (csrc/ctx.hip): not one copy of the history, no reallocation
    spike; the row-major mirror is given back once it would take more than its share.  Same checks as the 131 072-particle shard:
    the reference's N = 4096 twin walks 31 iterations to logZ -639.44 +- 0.09 and a posterior mean of v = 8.55 (the annealing
    schedule is set by ESS ratios, which do not depend on N).  profiles/r05_c5_full_2097152.json: 55.7 s, 153 GB at the peak."""
    import tempest_amd as tp
    dev = torch.device("cuda", 0)
    d, n = 100, 2097152
    free, total = torch.cuda.mem_get_info(dev)
    if free < 200 * 2 ** 30:
        pytest.skip(f"needs ~155 GB of device memory, {free / 2 ** 30:.0f} GB free")
    scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
    shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

    def loglike(x):
        v = x[:, 0]
        lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
        lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
        return lv + lr
    s = tp.Sampler(lambda u: u * scale + shift, loglike, d, n_particles=n, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True)
    t0 = time.time()
    s.run(n_total=4 * n, progress=False)
    torch.cuda.synchronize(dev)
    wall = time.time() - t0
    st = s.state
    ctx = st.ctx
    logz = s.evidence()[0]
    beta = np.asarray(st.get_history("beta")); steps = np.asarray(st.get_history("steps"))
    # the weighted mean of v = 30 u_0 - 15 over the whole history, on the device (10^8 rows never cross PCIe)
    m, s1, _ = st.reweight_eval([1.0])[0]
    mean_v = 30.0 * float(ctx.weighted_moments(ctx.weights(1.0, m, s1))[0].item()) - 15.0
    mem = ctx.history_memory()
    mu, sd, runs = _ref("c5twin_funnel100_n4096")
    ref_it = np.mean([r["iters"] for r in runs])
    print(f"config5 FULL to beta=1: logZ={logz:.3f} (twin reference {mu:.2f} +- {sd:.2f}) iters={len(beta)} (reference twin {ref_it:.0f}) "
          f"steps={int(steps[beta > 0].sum())} mean v={mean_v:.3f} wall={wall:.1f}s pms/s={steps[beta > 0].sum() * n / wall:.3g} memory={mem}")
    assert beta[-1] == 1.0 and np.all(np.diff(beta) >= 0) and wall < 150.0
    assert abs(len(beta) - ref_it) <= 3
    assert abs(mean_v - np.mean([r["mean"][0] for r in runs])) < 0.4
    assert abs(logz - mu) < 1.5
    assert mem["rows"] == n * len(beta) and mem["mapped"] == 1 and mem["copies"] == 0      # grown in place, never copied
    del s
    torch.cuda.empty_cache()
