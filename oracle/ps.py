"""NumPy restatement of the reference's persistent-SMC numerics (pure functions).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Each function cites the
file:line of minaskar/tempest v0.2.1 it follows.  All randomness is an explicit
argument (recorded draws or the Philox stream of ``oracle/philox.py``); nothing
here touches NumPy's global RNG.
"""
import numpy as np

SQRTEPS = float(np.sqrt(np.finfo(np.float64).eps))

# tempest/config.py:232-242
BETA_TOLERANCE = 1e-4
BETA_RTOL = 1e-8
ESS_TOLERANCE = 0.01
METRIC_ATOL = 0.5
METRIC_ATOL_CV = 0.01
DOF_FALLBACK = 1e6
TRIM_ESS = 0.99
TRIM_BINS = 1000


# ---------------------------------------------------------------- reweighting
def logaddexp_reduce_rows(b):
    """np.logaddexp.reduce(b, axis=1) written as the sequential fold it is."""
    acc = b[:, 0].copy()
    for t in range(1, b.shape[1]):
        acc = np.logaddexp(acc, b[:, t])
    return acc


def log_mixture(logl, beta_t, logz_t, n_t):
    """C_s = log sum_t n_t exp(beta_t l_s - logZ_t)   (state_manager.py:466-471 without
    the common -log N_h, which the device keeps out of the cached array)."""
    logl = np.asarray(logl, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        b = logl[:, None] * np.asarray(beta_t)[None, :] - np.asarray(logz_t)[None, :] \
            + np.log(np.asarray(n_t, dtype=np.float64))[None, :]
        return logaddexp_reduce_rows(b)


def compute_logw_and_logz(logl_all, beta_t, logz_t, n_t, beta_final=1.0, normalize=True):
    """state_manager.py:418-480."""
    beta_t = np.asarray(beta_t, dtype=np.float64)
    if beta_t.size == 0:
        return np.array([]), -np.inf
    logl_all = np.asarray(logl_all, dtype=np.float64)
    n_t = np.asarray(n_t)
    N_total = n_t.sum()
    with np.errstate(invalid="ignore"):
        A = logl_all * beta_final
    with np.errstate(invalid="ignore"):
        b = logl_all[:, None] * beta_t[None, :] - np.asarray(logz_t)[None, :]
        b_weighted = b + (np.log(n_t) - np.log(N_total))[None, :]
        B = np.logaddexp.reduce(b_weighted, axis=1)
        logw = A - B
        logz_new = np.logaddexp.reduce(logw) - np.log(logw.size)
        if normalize and logw.size:
            logw = logw - np.logaddexp.reduce(logw)
    return logw, logz_new


def reweight_triple(logl_all, cmix, beta):
    """(max v, sum e^{v-max}, sum e^{2(v-max)}) with v = beta*l - C: the device's K2 output."""
    with np.errstate(invalid="ignore"):
        v = beta * np.asarray(logl_all) - np.asarray(cmix)
    m = np.max(v)
    e = np.exp(v - m)
    return m, e.sum(), (e * e).sum()


def effective_sample_size(weights):
    """tools.py:120-135."""
    weights = weights / np.sum(weights)
    return 1.0 / np.sum(weights ** 2.0)


def increment_logz(logw):
    """tools.py:159-175."""
    logw_max = np.max(logw)
    return logw_max + np.logaddexp.reduce(logw - logw_max)


def trim_weights(samples, weights, ess=0.99, bins=1000):
    """tools.py:10-55 (does not modify its input, unlike the reference's in-place `/=`)."""
    weights = weights / np.sum(weights)
    ess_total = 1.0 / np.sum(weights ** 2.0)
    percentiles = np.linspace(0, 99, bins)
    i = bins - 1
    while True:
        threshold = np.percentile(weights, percentiles[i])
        mask = weights >= threshold
        wt = weights[mask]
        wt = wt / np.sum(wt)
        if (1.0 / np.sum(wt ** 2.0)) / ess_total >= ess:
            break
        i -= 1
    return samples[mask], wt


def trim_threshold_sorted(weights, ess=0.99, bins=1000, normalized=False):
    """The same decision as trim_weights taken the way the device takes it: one sort,
    suffix sums, all `bins` candidates at once.  Returns (threshold, kept_sum, kept_count)."""
    w = np.asarray(weights, dtype=np.float64)
    if not normalized:
        w = w / w.sum()
    s = np.sort(w)
    n = s.size
    t1 = np.cumsum(s[::-1])[::-1]
    t2 = np.cumsum((s * s)[::-1])[::-1]
    ess_total = t1[0] ** 2 / t2[0]
    p = np.linspace(0, 99, bins)
    thr = np.percentile(w, p)
    k = np.searchsorted(s, thr, side="left")
    ratio = (t1[k] ** 2 / t2[k]) / ess_total
    ok = np.nonzero(ratio >= ess)[0]
    i = ok.max()
    return thr[i], t1[k[i]], n - k[i]


def volume_variation(x, w=None):
    """tools.py:58-117."""
    x = np.asarray(x)
    n_samples, n_dim = x.shape
    if n_samples < n_dim + 1:
        return 1e10
    if w is None:
        w = np.ones(n_samples)
    w = np.asarray(w)
    w = w / np.sum(w)
    weighted_mean = np.sum(x * w[:, None], axis=0)
    xc = x - weighted_mean
    cov = np.dot(xc.T, xc * w[:, None])
    if np.linalg.matrix_rank(cov) < n_dim:
        cov = cov + np.eye(n_dim) * (1e-6 * np.trace(cov))
    try:
        cov_inv = np.linalg.inv(cov)
    except np.linalg.LinAlgError:
        return 1e10
    d2 = np.sum(xc @ cov_inv * xc, axis=1)
    deviation = np.clip(d2 - n_dim, -1e6, 1e6)
    return 0.5 * np.sqrt(np.sum(w ** 2 * deviation ** 2))


def find_ess_bracket(ess_fn, beta_current, ess_target,
                     beta_rtol=BETA_RTOL, beta_tol=BETA_TOLERANCE, trace=None):
    """steps/reweight.py:225-297.  ess_fn(beta)->ESS.  `trace` collects the trial betas."""
    def f(b):
        if trace is not None:
            trace.append(b)
        return ess_fn(b)
    beta_low, beta_high = beta_current, 1.0
    if f(beta_current) <= ess_target:
        return beta_current, beta_current
    if f(1.0) >= ess_target:
        return 1.0, 1.0
    while True:
        beta_mid = (beta_high + beta_low) * 0.5
        interval = beta_high - beta_low
        scale = max(abs(beta_low), abs(beta_high), np.finfo(float).tiny)
        if interval <= max(beta_rtol * scale, beta_tol * scale):
            break
        if f(beta_mid) >= ess_target:
            beta_low = beta_mid
        else:
            beta_high = beta_mid
    return beta_low, beta_high


def find_beta_bisection(metric_fn, beta_min, beta_max, target, dynamic=False,
                        ess_tol=ESS_TOLERANCE, beta_rtol=BETA_RTOL, beta_tol=BETA_TOLERANCE,
                        metric_atol=METRIC_ATOL, metric_atol_cv=METRIC_ATOL_CV, trace=None):
    """steps/reweight.py:123-223.  metric_fn(beta)->(metric, aux)."""
    beta, aux = None, None
    for _ in range(200):
        beta = (beta_max + beta_min) * 0.5
        if trace is not None:
            trace.append(beta)
        m, aux = metric_fn(beta)
        if not np.isfinite(m):
            m = 1e10
        atol = metric_atol_cv if dynamic else metric_atol
        metric_converged = abs(m - target) < max(ess_tol * abs(target), atol)
        scale = max(abs(beta_min), abs(beta_max), np.finfo(float).tiny)
        beta_converged = (beta_max - beta_min) < max(beta_rtol * scale, beta_tol * scale)
        if metric_converged or beta_converged or beta == 1.0:
            return beta, aux
        if not dynamic:
            if m < target:
                beta_max = beta
            else:
                beta_min = beta
        else:
            if m < target:
                beta_min = beta
            else:
                beta_max = beta
    return beta, aux


def reweighter_run(logl_all, u_all, beta_t, logz_t, n_t, beta_prev, n_particles,
                   ess_ratio=2.0, vol_var=None, trace=None):
    """steps/reweight.py:341-495 on an explicit history.
    Returns (beta, weights_normalised, ess, logz, cv)."""
    if len(beta_t) == 0:
        return 0.0, np.ones(n_particles) / n_particles, ess_ratio * n_particles, 0.0, 0.0

    def metric_and_weights(beta):
        logw, _ = compute_logw_and_logz(logl_all, beta_t, logz_t, n_t, beta)
        w = np.exp(logw - np.max(logw))
        ess = effective_sample_size(w)
        if vol_var is not None:
            metric = volume_variation(u_all, w / np.sum(w))
        else:
            metric = ess
        return w, ess, metric

    target = ess_ratio * n_particles
    beta_low, beta_high = find_ess_bracket(lambda b: metric_and_weights(b)[1], beta_prev, target,
                                           trace=trace)
    if vol_var is None:
        if beta_low == beta_high:
            beta = beta_low
            if trace is not None:
                trace.append(beta)
            w, ess, _ = metric_and_weights(beta)
        else:
            def ess_fn(b):
                w, e, _ = metric_and_weights(b)
                return e, (w, e)
            beta, (w, ess) = find_beta_bisection(ess_fn, beta_prev, beta_high, target, trace=trace)
    else:
        if beta_low == beta_high:
            beta = beta_low
            w, ess, _ = metric_and_weights(beta)
        else:
            _, ess_prev, vv_prev = metric_and_weights(beta_prev)
            _, ess_high, vv_high = metric_and_weights(beta_high)
            w = None
            if vol_var >= vv_high:
                beta, ess = beta_high, ess_high
            elif vol_var <= vv_prev:
                beta, ess = beta_prev, ess_prev
            else:
                def vv_fn(b):
                    w, e, m = metric_and_weights(b)
                    return m, (w, e)
                beta, (w, ess) = find_beta_bisection(vv_fn, beta_prev, beta_high, vol_var,
                                                     dynamic=True, trace=trace)
            if w is None:
                w, ess, _ = metric_and_weights(beta)
    cv = volume_variation(u_all, w / np.sum(w))
    _, logz = compute_logw_and_logz(logl_all, beta_t, logz_t, n_t, beta)
    return beta, w / np.sum(w), ess, logz, cv


# ----------------------------------------------------------------- resampling
def systematic_resample(size, weights, u0):
    """tools.py:178-228 with the single uniform `u0` made explicit; vectorised as
    idx_i = #{k : cumsum_k < pos_i} (the reference's strict `pos > cumsum` walk)."""
    weights = np.asarray(weights, dtype=np.float64)
    if abs(np.sum(weights) - 1.0) > SQRTEPS:
        weights = weights / np.sum(weights)
    positions = (u0 + np.arange(size)) / size
    cum = np.cumsum(weights)
    idx = np.searchsorted(cum, positions, side="left")
    return np.minimum(idx, weights.size - 1)


def multinomial_resample(weights, uniforms):
    """np.random.choice(n, size, p=w) as NumPy implements it (cdf, cdf/=cdf[-1],
    searchsorted side='right') with the uniforms made explicit (steps/resample.py:80-82)."""
    cdf = np.cumsum(np.asarray(weights, dtype=np.float64))
    cdf = cdf / cdf[-1]
    idx = np.searchsorted(cdf, uniforms, side="right")
    return np.minimum(idx, cdf.size - 1)


# ------------------------------------------------------------------- mutation
def apply_boundary_conditions(u, periodic=None, reflective=None):
    """mcmc.py:326-366."""
    u = np.array(u, dtype=np.float64, copy=True)
    if periodic is not None:
        for idx in periodic:
            u[..., idx] = u[..., idx] % 1.0
    if reflective is not None:
        for idx in reflective:
            val = u[..., idx]
            n_reflect = np.floor(val)
            remainder = val - n_reflect
            u[..., idx] = np.where(n_reflect % 2 == 0, remainder, 1.0 - remainder)
    return u


def check_bounds(u, periodic=None, reflective=None):
    """mcmc.py:369-411."""
    u = np.asarray(u)
    n_dim = u.shape[-1]
    special = set()
    if periodic is not None:
        special.update(int(i) for i in periodic)
    if reflective is not None:
        special.update(int(i) for i in reflective)
    strict = [i for i in range(n_dim) if i not in special]
    if len(strict) == 0:
        return True if u.ndim == 1 else np.ones(u.shape[0], dtype=bool)
    us = u[..., strict]
    if u.ndim == 1:
        return bool(np.all(us >= 0) and np.all(us <= 1))
    return np.all(us >= 0, axis=-1) & np.all(us <= 1, axis=-1)


def mahalanobis(u, means, inv_covs, assignments):
    """(u-mu)^T Sigma^-1 (u-mu) per particle (mcmc.py:233,258-261)."""
    diff = u - means[assignments]
    return np.einsum("ij,ijk,ik->i", diff, inv_covs[assignments], diff)


def tpcn_proposal(u, means, chol_covs, assignments, sigmas, s, z):
    """mcmc.py:240-244 for all walkers at once, given the Gamma-derived `s` and normals `z`."""
    mu = means[assignments]
    diff = u - mu
    sig = sigmas[assignments][:, None]
    Lz = np.einsum("ijk,ik->ij", chol_covs[assignments], z)
    return mu + np.sqrt(1.0 - sig ** 2.0) * diff + sig * np.sqrt(s)[:, None] * Lz


def rwm_proposal(u, chol_covs, assignments, sigmas, z):
    """mcmc.py:307."""
    sig = sigmas[assignments][:, None]
    return u + sig * np.einsum("ijk,ik->ij", chol_covs[assignments], z)


def tpcn_acceptance_factor(u, u_prime, means, inv_covs, dof, assignments):
    """mcmc.py:251-279:  -A + B."""
    n_dim = u.shape[1]
    nu = dof[assignments]
    B = -0.5 * (n_dim + nu) * np.log(1 + mahalanobis(u, means, inv_covs, assignments) / nu)
    A = -0.5 * (n_dim + nu) * np.log(1 + mahalanobis(u_prime, means, inv_covs, assignments) / nu)
    return -A + B


def metropolis_alpha(beta, logl, logl_prime, factor):
    """mcmc.py:163-166."""
    with np.errstate(over="ignore", invalid="ignore"):
        alpha = np.exp(beta * (logl_prime - logl) + factor)
    alpha = np.minimum(1.0, alpha)
    return np.nan_to_num(alpha, nan=0.0)


def adapt_sigma_tpcn(sigma, mean_accept, iteration, sigma_0):
    """mcmc.py:281-288."""
    return float(np.clip(sigma + (mean_accept - 0.234) / (iteration + 1), 0, min(sigma_0, 0.99)))


def adapt_sigma_rwm(sigma, mean_accept, iteration):
    """mcmc.py:320-323."""
    return sigma + (mean_accept - 0.234) / (iteration + 1)


def adaptive_steps(n_steps, n_max, n_dim, sigmas, assignments, n_clusters, current_acceptance):
    """mcmc.py:104-135."""
    sizes = np.array([np.sum(assignments == c) for c in range(n_clusters)])
    sizes = sizes[sizes > 0]
    weighted_sigma = np.average(sigmas[: len(sizes)], weights=sizes)
    sigma_0 = 2.38 / np.sqrt(n_dim)
    n_min = n_steps * n_dim
    n_adapt = n_steps * n_dim * (0.234 / max(0.01, current_acceptance)) \
        * (sigma_0 / max(1e-6, weighted_sigma)) ** 2
    return int(min(max(n_min, n_adapt), n_max * n_dim))


# ------------------------------------------------------------ proposal fitting
def fit_mvstud_effective(data):
    """What student.py:6-116 returns with NumPy 2.2 / SciPy 1.15: the first EM
    pass exits with nu=inf, so (per-dimension median, MLE covariance + diag(var)/n, inf);
    ridge only if Cholesky of Sigma fails (student.py:60-64,75-79)."""
    data = np.asarray(data, dtype=np.float64).T
    dim, n = data.shape
    mu = np.median(data, 1)
    Sigma = np.cov(data) * (n - 1) / n + (1 / n) * np.diag(np.var(data, axis=1))
    Sigma = np.atleast_2d(Sigma)
    try:
        np.linalg.cholesky(Sigma)
    except np.linalg.LinAlgError:
        Sigma = Sigma + np.eye(dim) * max(1e-6, 1e-6 * abs(np.trace(Sigma)))
    return mu, Sigma, np.inf


def median_cov_from_counts(u, counts):
    """fit_mvstud_effective on the multiset {u_s repeated counts_s times} without building it
    (how the device evaluates modes.py:196-205 after up-sampling)."""
    counts = np.asarray(counts, dtype=np.int64)
    keep = counts > 0
    return fit_mvstud_effective(np.repeat(u[keep], counts[keep], axis=0))


def mode_statistics(means, covariances):
    """modes.py:102-119: Cholesky + inverse per mode, ridge on LinAlgError."""
    means = np.atleast_2d(means)
    covariances = np.array(covariances, dtype=np.float64, copy=True)
    if covariances.ndim == 2:
        covariances = covariances[None]
    chol = np.empty_like(covariances)
    inv = np.empty_like(covariances)
    for k in range(means.shape[0]):
        c = covariances[k]
        try:
            chol[k] = np.linalg.cholesky(c)
            inv[k] = np.linalg.inv(c)
        except np.linalg.LinAlgError:
            c = c + np.eye(c.shape[0]) * max(1e-6, 1e-6 * abs(np.trace(c)))
            covariances[k] = c
            chol[k] = np.linalg.cholesky(c)
            inv[k] = np.linalg.inv(c)
    return covariances, chol, inv


def mode_stats_from_global(u, weights, upsample_idx, dof_fallback=DOF_FALLBACK):
    """modes.py:221-288 with the up-sampling indices made explicit."""
    mean, cov, dof = fit_mvstud_effective(u[upsample_idx])
    if not np.isfinite(dof):
        dof = dof_fallback
    cov, chol, inv = mode_statistics(mean.reshape(1, -1), cov.reshape(1, *cov.shape))
    return mean.reshape(1, -1), cov, chol, inv, np.array([dof])


def inf_repair(logl, choice_uniforms):
    """steps/mutate.py:122-148: rows with +-inf logl are replaced by uniformly chosen finite rows;
    returns (source index per row, log(n_finite/n)).  `choice_uniforms` has one U[0,1) per row
    (only the infinite rows consume theirs)."""
    logl = np.asarray(logl)
    n = logl.size
    src = np.arange(n)
    infm = np.isinf(logl)
    if not infm.any():
        return src, 0.0
    fin = np.nonzero(~infm)[0]
    if fin.size > 0:
        pick = np.minimum((choice_uniforms[infm] * fin.size).astype(np.int64), fin.size - 1)
        src[infm] = fin[pick]
    with np.errstate(divide="ignore"):
        return src, float(np.log(fin.size / n))


def fit_mvstud_em(data, counts=None, tolerance=1e-6, max_iter=100, nu_hi=1e6, nu_lo=1e-2):
    """The EM of tempest/student.py:66-116 WITH a working degrees-of-freedom update (opt-in extension).

    The reference evaluates the digamma equation (student.py:40-57) at nu = 1e300, where it rounds to a value >= 0 for any
    data: nu becomes inf on the first pass and the start values are returned (`fit_mvstud_effective` above).  Here the loop
    is the reference's line by line -- start values (:60-64), ridge on a failed Cholesky (:73-77), delta_i (:79-87), nu from
    the root of func0 (:40-57), Sigma = sum_i w_i diff_i diff_i^T / n about the OLD mean (:96-98), mu = sum w x / sum w
    (:100-102), stop on |nu - last_nu| <= tolerance -- with ONE change: the root is bracketed on [nu_lo, nu_hi] = [1e-2, 1e6]
    (func0(nu_hi) >= 0 -> nu = inf: the Gaussian limit; func0(nu_lo) <= 0 -> nu_lo).  `counts`: integer multiplicities of the
    rows (the x4 up-sampling of modes.py:196-201 as counts); n = sum of the counts.  PARITY UNPINNED by the reference (its
    own loop never iterates): this restatement is the checker of the device version."""
    from scipy import optimize, special
    X = np.asarray(data, dtype=np.float64)
    n_rows, dim = X.shape
    c = np.ones(n_rows) if counts is None else np.asarray(counts, dtype=np.float64)
    n = c.sum()
    mu, Sigma, _ = fit_mvstud_effective(np.repeat(X, c.astype(int), axis=0)) if counts is not None else fit_mvstud_effective(X)
    nu, last_nu, it = 20.0, 0.0, 0

    def ridge(S):
        try:
            np.linalg.cholesky(S)
            return S
        except np.linalg.LinAlgError:
            return S + np.eye(dim) * max(1e-6, 1e-6 * abs(np.trace(S)))
    while abs(last_nu - nu) > tolerance and it < max_iter:
        it += 1
        Sigma = ridge(Sigma)
        diffs = X - mu
        delta = np.sum(diffs * np.linalg.solve(Sigma, diffs.T).T, axis=1)

        def func0(v):
            w = (v + dim) / (v + delta)
            return (-special.psi(v / 2) + np.log(v / 2) + np.sum(c * np.log(w)) / n - np.sum(c * w) / n + 1
                    + special.psi((v + dim) / 2) - np.log((v + dim) / 2))
        last_nu = nu
        if func0(nu_hi) >= 0:
            return mu, Sigma, np.inf, it
        nu = nu_lo if func0(nu_lo) <= 0 else optimize.brentq(func0, nu_lo, nu_hi, xtol=1e-12, rtol=1e-13)
        w = c * (nu + dim) / (nu + delta)
        Sigma = (diffs * w[:, None]).T @ diffs / n
        mu = (w[:, None] * X).sum(axis=0) / w.sum()
    return mu, ridge(Sigma), nu, it

