"""The Student-t EM (opt-in extension of tempest/student.py:66-116 with a working degrees-of-freedom update) on the device,
against its NumPy restatement (oracle.ps.fit_mvstud_em).  PARITY UNPINNED BY THE REFERENCE: with the NumPy / SciPy versions the
reference pins its own loop leaves at the first pass with nu = inf (the default `fit_mvstud` reproduces that, G7) --
there is no reference output for the iterating EM.  Run on the GPU box:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import ps  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _mvt(rs, n, d, nu):
    A = rs.randn(d, d) / np.sqrt(d)
    S = A @ A.T + 0.3 * np.eye(d)
    L = np.linalg.cholesky(S)
    g = rs.chisquare(nu, size=n) / nu
    return 0.5 + (rs.randn(n, d) @ L.T) / np.sqrt(g)[:, None] * 0.05, S * 0.05 ** 2


@pytest.mark.parametrize("d,n,nu_true", [(3, 20000, 4.0), (10, 50000, 8.0), (32, 40000, 3.0)])
def test_student_em_equals_the_restatement_and_recovers_the_degrees_of_freedom(d, n, nu_true):
    from tempest_amd.student import fit_mvstud
    rs = np.random.RandomState(100 + d)
    X, S = _mvt(rs, n, d, nu_true)
    mu, Sigma, nu = fit_mvstud(X, em=True)
    omu, oSigma, onu, its = ps.fit_mvstud_em(X)
    assert its > 2 and np.isfinite(onu)
    np.testing.assert_allclose(nu, onu, rtol=1e-6)
    np.testing.assert_allclose(mu, omu, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Sigma, oSigma, rtol=1e-7, atol=1e-12)
    assert abs(nu - nu_true) < 0.2 * nu_true                  # the EM does recover the tail index
    np.testing.assert_allclose(Sigma, S, rtol=0.15, atol=0.1 * np.abs(S).max())      # and the scale matrix of the t law
    # the default is the reference's effective estimator, untouched: start values, nu = inf
    m0, S0, nu0 = fit_mvstud(X)
    em0, eS0, _ = ps.fit_mvstud_effective(X)
    assert nu0 == np.inf
    np.testing.assert_allclose(m0, em0, rtol=1e-12)
    np.testing.assert_allclose(S0, eS0, rtol=1e-10)


def test_student_em_on_gaussian_data_goes_to_the_gaussian_limit():
    from tempest_amd.student import fit_mvstud
    rs = np.random.RandomState(5)
    X = 0.5 + 0.1 * rs.randn(30000, 6)
    mu, Sigma, nu = fit_mvstud(X, em=True)
    omu, oSigma, onu, _ = ps.fit_mvstud_em(X)
    assert (nu == np.inf and onu == np.inf) or (nu > 50 and onu > 50 and abs(nu - onu) < 1e-4 * onu)
    np.testing.assert_allclose(mu, omu, rtol=1e-8)
    np.testing.assert_allclose(Sigma, oSigma, rtol=1e-7)


def test_sampler_with_student_em_runs_and_reports_finite_dof():
    """Sampler(student_em=True) end to end on a heavy-tailed target: the proposal's degrees of freedom are finite and the
    evidence stays within the usual window of the analytic value."""
    import tempest_amd as tp
    d = 6
    dev = torch.device("cuda", 0)
    nu_t = 4.0
    from scipy.special import gammaln
    const = float(gammaln((nu_t + d) / 2) - gammaln(nu_t / 2) - 0.5 * d * np.log(nu_t * np.pi))

    def loglike(x):                      # standard multivariate t, nu = 4
        return const - 0.5 * (nu_t + d) * torch.log1p((x * x).sum(dim=1) / nu_t)
    s = tp.Sampler(lambda u: 60 * u - 30, loglike, d, n_particles=2048, vectorize=True, clustering=False, random_state=3, student_em=True)
    seen = []
    from tempest_amd.steps import train as tr
    orig = tr.Trainer.run

    def run(self, w):
        ms = orig(self, w)
        seen.append(float(ms.degrees_of_freedom[0]))
        return ms
    tr.Trainer.run = run
    try:
        s.run(n_total=8192, progress=False)
    finally:
        tr.Trainer.run = orig
    late = [v for v in seen[-5:]]
    print("dof by iteration:", [round(v, 2) for v in seen], "logZ", s.evidence()[0], "analytic", -d * np.log(60.0))
    assert all(np.isfinite(v) for v in late) and min(late) > 1.0 and max(late) < 50.0
    assert abs(s.evidence()[0] + d * np.log(60.0)) < 0.5
