"""End-to-end parity of the GPU sampler with the reference (run on the GPU box: pytest -m gpu).

The reference's RNG cannot be reproduced, so parity is statistical: the reference was run
here over 16 seeds per configuration (oracle/make_ref_ensembles.py -> tests/golden/ref_ensembles.json);
each GPU run must land within 3 sigma_ref of the reference ensemble mean, and the
reference's own end-to-end acceptance test (tests/test_end_to_end.py:31-76) must pass unchanged."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

G = os.path.join(os.path.dirname(__file__), "golden")


def ref_stats(name):
    runs = [r for r in json.load(open(os.path.join(G, "ref_ensembles.json")))["runs"] if r["config"] == name]
    lz = np.array([r["logz"] for r in runs])
    return lz.mean(), lz.std(ddof=1), runs


E2E_MEAN = np.array([2.0, -1.5, 0.5, 3.2, -2.8, 1.1, -0.7, 2.5, -1.2, 0.9])
E2E_VAR = np.array([1.0, 0.8, 1.2, 0.9, 1.1, 0.7, 1.3, 0.85, 1.15, 0.95])


def prior20(u):
    return 20 * u - 10


def make_gauss(dev):
    mean = torch.from_numpy(E2E_MEAN).to(dev)
    var = torch.from_numpy(E2E_VAR).to(dev)
    const = float(-0.5 * np.sum(np.log(2 * np.pi * E2E_VAR)))

    def loglike(x):
        return -0.5 * ((x - mean) ** 2 / var).sum(dim=1) + const
    return loglike


def rosenbrock(x):
    return -(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0).sum(dim=1)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def test_reference_end_to_end_acceptance(dev):
    """reference tests/test_end_to_end.py:31-76 with the sampler swapped."""
    import tempest_amd as tp
    s = tp.Sampler(prior_transform=prior20, log_likelihood=make_gauss(dev), n_dim=10, vectorize=True,
                   n_particles=128, clustering=False, random_state=42, n_steps=1)
    s.run(n_total=2048, progress=False)
    x, w, logl = s.posterior(trim_importance_weights=True, resample=False)
    assert x.flags["C_CONTIGUOUS"] and x.shape[1] == 10 and w.shape == logl.shape == (x.shape[0],)
    np.testing.assert_allclose(w.sum(), 1.0, rtol=1e-12)
    mean = np.average(x, weights=w, axis=0)
    cov = np.cov(x, rowvar=False, aweights=w)
    np.testing.assert_allclose(mean, E2E_MEAN, atol=0.25, rtol=0)
    np.testing.assert_allclose(np.diag(cov), E2E_VAR, atol=0.5, rtol=0)
    logz, err = s.evidence()
    assert err is None and abs(logz - (-29.96)) <= 0.5, logz
    assert s.state.get_current("beta") > 0.99
    assert s.state.get_current("acceptance") > 0.1
    # bookkeeping identities of the reference (mcmc.py:89, mutate.py:106,199)
    beta = np.asarray(s.state.get_history("beta")); steps = np.asarray(s.state.get_history("steps"))
    assert s.state.get_current("calls") == 128 * (np.sum(beta == 0) + np.sum(steps[beta > 0]))
    assert np.sum(beta == 0) == 3                     # warm-up length with ess_ratio = 2
    assert s.state.get_history("u", flat=True).shape == (128 * len(beta), 10)


@pytest.mark.parametrize("cfg,kw", [("e2e_gauss10_n128", dict(n_steps=1)),
                                    ("e2e_gauss10_n128_rwm_syst", dict(sample="rwm", resample="syst")),
                                    # dynamic beta schedule (volume-variation target, reweight.py:427-495) end to end
                                    ("e2e_gauss10_n128_dynamic", dict(volume_variation=0.5)),
                                    # boundary conditions inside the proposal (mcmc.py:326-411) end to end
                                    ("e2e_gauss10_n128_bc", dict(periodic=[0, 3], reflective=[1, 4]))])
def test_logz_within_3sigma_of_reference_gauss(dev, cfg, kw):
    import tempest_amd as tp
    mu, sd, runs = ref_stats(cfg)
    got = []
    for seed in range(8):
        s = tp.Sampler(prior20, make_gauss(dev), 10, vectorize=True, n_particles=128, clustering=False,
                       random_state=seed, **kw)
        s.run(n_total=2048, progress=False)
        got.append(s.evidence()[0])
    got = np.array(got)
    print(cfg, "ref", mu, sd, "gpu", got.mean(), got.std(ddof=1))
    assert np.all(np.abs(got - mu) <= 3 * sd + 0.1), (got, mu, sd)
    assert abs(got.mean() - mu) <= 3 * sd * np.sqrt(1.0 / len(got) + 1.0 / len(runs)) + 0.05, (got.mean(), mu, sd)
    z = (got.mean() - mu) / np.sqrt((sd ** 2 + got.var(ddof=1)) / len(got))
    assert abs(z) < 4.0, z
    ref_it = np.mean([r["iters"] for r in runs])
    assert abs(len(s.state.get_history("beta")) - ref_it) < 6


def test_rosenbrock_config1_parity(dev):
    """BASELINE config 1 (README Rosenbrock, N=1000, clustering=False twin): every run's logZ within 3 sigma_ref (+0.1, the
    allowance the other ensemble gates carry: a run of this build is a draw from a distribution of the same width, and a
    fixed seed list is a new realisation whenever the counter-based stream's layout changes) of the reference ensemble mean,
    the ensemble mean within its standard error of it (16 seeds of this build: -29.755 +- 0.033 against -29.804 +- 0.029),
    posterior moments near the reference's and the analytic ones."""
    import tempest_amd as tp
    mu, sd, runs = ref_stats("c1_rosenbrock_nocluster")
    ref_mean = np.mean([r["mean"] for r in runs], axis=0)
    ref_var = np.mean([r["var"] for r in runs], axis=0)
    got, means, vars_ = [], [], []
    for seed in range(6):
        s = tp.Sampler(prior20, rosenbrock, 10, vectorize=True, n_particles=1000, clustering=False, random_state=seed)
        s.run(progress=False)
        got.append(s.evidence()[0])
        x, w, _ = s.posterior()
        m = np.average(x, weights=w, axis=0)
        means.append(m); vars_.append(np.average((x - m) ** 2, weights=w, axis=0))
    got = np.array(got)
    print("rosenbrock ref", mu, sd, "gpu", got, "analytic", -29.9901)
    assert np.all(np.abs(got - mu) <= 3 * sd + 0.1), (got, mu, sd)
    assert abs(got.mean() - mu) <= 3 * sd * np.sqrt(1.0 / len(got) + 1.0 / len(runs)) + 0.05, (got.mean(), mu, sd)
    np.testing.assert_allclose(np.mean(means, axis=0), ref_mean, atol=0.15)
    np.testing.assert_allclose(np.mean(vars_, axis=0), ref_var, rtol=0.25)
    pms = np.sum(np.asarray(s.state.get_history("steps"))[np.asarray(s.state.get_history("beta")) > 0]) * 1000
    ref_pms = np.mean([r["pms"] for r in runs])
    assert 0.5 * ref_pms < pms < 2.0 * ref_pms, (pms, ref_pms)


def test_same_seed_same_result_and_numpy_backend(dev):
    import tempest_amd as tp

    def run(backend, ll):
        s = tp.Sampler(prior20, ll, 4, vectorize=True, n_particles=64, clustering=False, random_state=5,
                       backend=backend)
        s.run(n_total=256, progress=False)
        return s

    tl = lambda x: -0.5 * (x ** 2).sum(dim=1)          # noqa: E731
    nl = lambda x: -0.5 * np.sum(x ** 2, axis=1)       # noqa: E731
    a, b, c = run("torch", tl), run("torch", tl), run("auto", nl)
    assert a.evidence()[0] == b.evidence()[0]          # bitwise reproducible
    np.testing.assert_array_equal(a.state.get_history("x", flat=True), b.state.get_history("x", flat=True))
    assert c._core.callbacks.backend == "numpy"        # NumPy likelihood detected, staged through the host
    assert abs(c.evidence()[0] - a.evidence()[0]) < 1e-6
    assert abs(a.evidence()[0] - (4 * np.log(np.sqrt(2 * np.pi) / 20))) < 0.5


def test_periodic_reflective_and_state_api(dev):
    import tempest_amd as tp
    s = tp.Sampler(prior20, lambda x: -0.5 * (x ** 2).sum(dim=1), 3, vectorize=True, n_particles=32,
                   clustering=False, periodic=[0], reflective=[1], random_state=1)
    out = s.sample()
    assert set(out) >= {"u", "x", "logl", "beta", "logz", "iter", "calls"}
    assert out["u"].shape == (32, 3) and out["iter"] == 1 and out["beta"] == 0.0
    out["u"][:] = -1            # returned arrays are copies (reference tests/test_sample_method.py:125-146)
    assert s.state.get_current("u").min() >= 0
    s.run(n_total=128, progress=False)
    u = s.state.get_history("u", flat=True)
    assert u.min() >= 0 and u.max() <= 1
    with pytest.raises(ValueError):
        s.state.get_current("nope")
    with pytest.raises(IndexError):
        s.state.get_history("beta", index=10_000)
    r = s.results()
    assert "logw" in r and len(r["logw"]) == u.shape[0]
    np.testing.assert_allclose(np.exp(r["logw"]).sum(), 1.0, rtol=1e-10)


@pytest.mark.parametrize("kw", [dict(sample="tpcn", resample="mult"), dict(sample="rwm", resample="syst"),
                                dict(sample="tpcn", resample="syst", periodic=[0], reflective=[2])])
def test_whole_run_matches_oracle_on_the_same_seed(dev, kw):
    """The device sampler and the NumPy oracle sampler consume the same counter-based stream in the same order, so a
    whole run (warm-up, beta schedule, proposal fits, resampling, every MCMC step) can be compared number by number:
    identical beta / step schedules and the same evidence to ~1e-9."""
    import tempest_amd as tp
    from oracle.sampler import OracleSampler
    d, n = 6, 256
    mean_h = np.linspace(-2, 2, d)
    mean_t = torch.from_numpy(mean_h).to(dev)
    const = -0.5 * d * np.log(2 * np.pi)
    s = tp.Sampler(prior20, lambda x: -0.5 * ((x - mean_t) ** 2).sum(dim=1) + const, d, n_particles=n, vectorize=True,
                   clustering=False, random_state=11, **kw)
    s.run(n_total=1024, progress=False)
    o = OracleSampler(prior20, lambda x: -0.5 * np.sum((x - mean_h) ** 2, axis=1) + const, d, n, seed=11,
                      sample=kw["sample"], resample=kw["resample"], periodic=kw.get("periodic"),
                      reflective=kw.get("reflective"))
    oz = o.run(1024)
    beta = np.asarray(s.state.get_history("beta"))
    assert len(beta) == len(o.hist["beta"])
    np.testing.assert_allclose(beta, o.hist["beta"], rtol=1e-7, atol=1e-12)
    np.testing.assert_array_equal(np.asarray(s.state.get_history("steps")), o.hist["steps"])
    np.testing.assert_allclose(np.asarray(s.state.get_history("logz")), o.hist["logz"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.asarray(s.state.get_history("acceptance")), o.hist["acceptance"], rtol=1e-6)
    assert abs(s.evidence()[0] - oz) < 1e-6, (s.evidence()[0], oz)
    assert s.state.get_current("calls") == o.cur["calls"]
    # the particles themselves agree to rounding
    np.testing.assert_allclose(s.state.get_history("x", index=len(beta) - 1), o.hist["x"][-1], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_logz_within_3sigma_of_reference_gauss20(dev, kernel):
    """20-D correlated Gaussian (Sigma = A A^T/20 + 0.5 I, A from RandomState(1)), N=256: the reference's own ensembles
    sit at -59.28 +- 0.14 (tpCN) and -57.71 +- 0.26 (RWM) against the analytic -59.91 (the algorithm's evidence excess
    grows with dimension and is larger for the less efficient kernel); the GPU runs must land in the same place."""
    import tempest_amd as tp
    mu, sd, runs = ref_stats("gauss20_n256_" + kernel)
    d = 20
    A = np.random.RandomState(1).randn(d, d)
    S = A @ A.T / d + 0.5 * np.eye(d)
    P = torch.from_numpy(np.linalg.inv(S)).to(dev)
    const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
    got = []
    for seed in range(8):
        s = tp.Sampler(prior20, lambda x: -0.5 * ((x @ P) * x).sum(dim=1) + const, d, vectorize=True, n_particles=256,
                       clustering=False, random_state=seed, sample=kernel)
        s.run(n_total=1024, progress=False)
        got.append(s.evidence()[0])
    got = np.array(got)
    print("gauss20", kernel, "ref", mu, sd, "gpu", got.mean(), got.std(ddof=1), "analytic", -d * np.log(20.0))
    assert np.all(np.abs(got - mu) <= 3 * sd + 0.1), (got, mu, sd)
    assert abs(got.mean() - mu) < 3 * sd / np.sqrt(8) + 0.1
    assert abs(len(s.state.get_history("beta")) - np.mean([r["iters"] for r in runs])) < 4
