"""Size-independent properties of the hot-path kernels at BASELINE.json's full single-GPU sizes
(config 4 shard: 131 072 particles x 40 iterations = 5.2e6 history rows; config 2/3: 2.6e6 / 1.05e7 rows).
The oracle cannot run these sizes in seconds, so the checks are invariants of the domain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def big():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tempest_amd.device import HipContext
    d, n, T = 10, 262144, 40                      # 1.05e7 rows (config 3's history length)
    dev = torch.device("cuda", 0)
    c = HipContext(d, 0, n * T)
    g = torch.Generator(device=dev).manual_seed(1)
    betas = np.concatenate([[0, 0, 0], np.linspace(0.001, 1.0, T - 3) ** 2])
    logz = -np.concatenate([[0, 0, 0], np.linspace(1, 30, T - 3)])
    for t in range(T):
        u = torch.rand(d, n, dtype=torch.float64, device=dev, generator=g)
        x = 20 * u - 10
        logl = -0.5 * (x ** 2).sum(dim=0) * (1.0 - 0.9 * betas[t])
        c.history_append(u, x, logl.contiguous(), betas[t], logz[t])
    return c, dev, d, n, T


def test_reweight_invariants_fullsize(big):
    c, dev, d, n, T = big
    nh = c.size
    assert nh == n * T
    betas = np.linspace(0.0, 1.0, 16)
    tri = c.reweight_eval(betas)
    ess = tri[:, 1] ** 2 / tri[:, 2]
    assert np.all(ess > 0) and np.all(ess <= nh * (1 + 1e-12))
    # weights: normalised, ESS consistent with the triple, a batch equals single evaluations bitwise
    for j in (3, 15):
        w = c.weights(betas[j], tri[j, 0], tri[j, 1])
        s = c.sum_sq_max(w)
        np.testing.assert_allclose(s[0], 1.0, rtol=1e-11)
        np.testing.assert_allclose(1.0 / s[1], ess[j], rtol=1e-10)
        one = c.reweight_eval([betas[j]])[0]
        np.testing.assert_allclose(one[0] + np.log(one[1]), tri[j, 0] + np.log(tri[j, 1]), rtol=1e-13)
    # linearity of the evidence in a constant shift of the log-likelihood is exercised through logw:
    lw = c.logw(1.0, nh)
    np.testing.assert_allclose(torch.logsumexp(lw, 0).item() - np.log(nh), tri[15, 0] + np.log(tri[15, 1]), rtol=1e-12)


def test_resampling_invariants_fullsize(big):
    c, dev, d, n, T = big
    tri = c.reweight_eval([0.5])[0]
    w = c.weights(0.5, tri[0], tri[1])
    cdf = c.cdf(w)
    # a parallel FP64 scan is monotone up to rounding: neighbouring partial sums come from different summation trees
    assert float(((cdf[1:] - cdf[:-1]) / cdf[1:]).min()) > -1e-15 and abs(cdf[-1].item() - 1.0) < 1e-10
    # systematic: sorted output, every row's count within 1 of N*w (stratified), encode -> gather round trip
    idx = c.resample_systematic(cdf, n, 0.37)
    assert bool((idx[1:] >= idx[:-1]).all()) and 0 <= int(idx.min()) and int(idx.max()) < c.size
    cnt = torch.bincount(idx, minlength=c.size).to(torch.float64)
    assert float((cnt - n * w).abs().max()) <= 1.0 + 1e-9
    uo, xo, lo = c.empty(d, n), c.empty(d, n), c.empty(n)
    c.gather(idx, uo, xo, lo)
    assert torch.equal(xo, 20 * uo - 10)                                   # rows stay rows
    sel = idx[:: n // 64].cpu().numpy()
    from tempest_amd.device import KEY_U
    for k, s in zip(range(0, n, n // 64), sel):
        np.testing.assert_array_equal(uo[:, k].cpu().numpy(), c.history_read(KEY_U, int(s), 1)[0])
    # multinomial: in range, mean index weight matches, deterministic in (seed, tick)
    a = c.resample_multinomial(cdf, n, 5, 9)
    b = c.resample_multinomial(cdf, n, 5, 9)
    assert torch.equal(a, b) and 0 <= int(a.min()) and int(a.max()) < c.size
    assert not torch.equal(a, c.resample_multinomial(cdf, n, 5, 10))
    top = torch.topk(w, 1000).indices
    hit = torch.isin(a, top).double().mean().item()
    np.testing.assert_allclose(hit, w[top].sum().item(), rtol=0.1)


def test_trim_and_fit_invariants_fullsize(big):
    c, dev, d, n, T = big
    tri = c.reweight_eval([1.0])[0]
    w = c.weights(1.0, tri[0], tri[1])
    thr, host = c.trim_threshold(w, 0.99, 1000, sync=True)
    kept = w >= thr[0]
    assert int(kept.sum()) == int(host[2])
    np.testing.assert_allclose(w[kept].sum().item(), host[1], rtol=1e-11)
    wk = w[kept] / host[1]
    ess_ratio = (1.0 / (wk ** 2).sum().item()) / host[3]
    assert ess_ratio >= 0.99 - 1e-12                                        # the defining property (tools.py:51)
    # idempotence: trimming the already trimmed, renormalised weights again keeps (almost) everything of its ESS
    cdf = c.cdf(w, thr[0:1])
    counts = c.multinomial_counts(cdf, 3, 4, kept_count=thr[2:3], factor=4, n_draw_max=4 * c.size)
    assert int(counts.sum()) == 4 * int(host[2]) and int(counts[~kept].sum()) == 0
    means, covs, chol, inv, _ = c.fit_modes(counts)
    L = chol[0]
    np.testing.assert_allclose((L @ L.T).cpu().numpy(), covs[0].cpu().numpy(), rtol=1e-10)
    np.testing.assert_allclose((covs[0] @ inv[0]).cpu().numpy(), np.eye(d), atol=1e-8)
    # the fitted median/covariance describe the weighted history: target is N(0, 1/(1-0.9 b)...) centred at u = 0.5
    np.testing.assert_allclose(means[0].cpu().numpy(), 0.5, atol=5e-3)
    assert bool((torch.diagonal(covs[0]) > 0).all())
