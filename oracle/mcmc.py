"""CPU twin of the device mutation kernels, draw-for-draw on the shared Philox stream.

TEST INFRASTRUCTURE ONLY.  Restates tempest/mcmc.py:142-323 (runner loop, tpCN / RWM proposals,
acceptance, sigma adaptation, adaptive step count) and tempest/steps/mutate.py:99-149 (beta=0 prior
draw + inf repair) for all walkers at once, with the reference's global NumPy RNG replaced by the
counter-based stream of oracle/philox.py.  Particle arrays here are (n, d) row-major
like the reference's.
"""
import numpy as np

from . import philox as px
from . import ps

MAX_ATTEMPTS = 256


def bc_flags(n_dim, periodic=None, reflective=None):
    f = np.zeros(n_dim, dtype=np.uint8)
    if periodic is not None:
        f[np.asarray(periodic, dtype=int)] = 1
    if reflective is not None:
        f[np.asarray(reflective, dtype=int)] = 2
    return f


def _apply_bc(v, flags):
    per = np.nonzero(flags == 1)[0]
    ref = np.nonzero(flags == 2)[0]
    v = ps.apply_boundary_conditions(v, per if per.size else None, ref if ref.size else None)
    strict = np.nonzero(flags == 0)[0]
    if strict.size:
        ok = np.all((v[:, strict] >= 0) & (v[:, strict] <= 1), axis=1)
    else:
        ok = np.ones(v.shape[0], dtype=bool)
    return v, ok


def propose(kernel, u, assign, means, chol, inv, dof, sigmas, flags, seed, tick, item0=0):
    """mcmc.py:225-249 (tpCN) / :301-312 (RWM) with redraw-until-in-bounds.
    Returns (u_prime, maha_u, maha_up)."""
    n, d = u.shape
    items = np.arange(n, dtype=np.uint64) + np.uint64(item0)
    sig = sigmas[assign]
    if kernel == "tpcn":
        mu = means[assign]
        diff = u - mu
        m_u = np.einsum("ij,ijk,ik->i", diff, inv[assign], diff)
        nu = dof[assign]
        gam = px.gamma_mt(seed, items, 0.5 * (d + nu), tick) * (2.0 / (nu + m_u))   # mcmc.py:234-236
        s = 1.0 / gam
        a_fac = np.sqrt(1.0 - sig * sig)
        b_fac = sig * np.sqrt(s)
    else:
        diff = u
        m_u = np.zeros(n)
        b_fac = sig
    out = np.array(u, copy=True)
    todo = np.ones(n, dtype=bool)
    for att in range(MAX_ATTEMPTS):
        if not todo.any():
            break
        idx = np.nonzero(todo)[0]
        z = px.normals(seed, items[idx], d, tick, px.TAG_NORMAL, attempt=att)
        Lz = np.einsum("ijk,ik->ij", chol[assign[idx]], z)
        if kernel == "tpcn":
            v = mu[idx] + a_fac[idx, None] * diff[idx] + b_fac[idx, None] * Lz
        else:
            v = diff[idx] + b_fac[idx, None] * Lz
        v, ok = _apply_bc(v, flags)
        out[idx[ok]] = v[ok]
        todo[idx[ok]] = False
    if kernel == "tpcn":
        dp = out - mu
        m_up = np.einsum("ij,ijk,ik->i", dp, inv[assign], dp)
    else:
        m_up = np.zeros(n)
    return out, m_u, m_up


def accept(kernel, beta, logl, logl_prime, maha_u, maha_up, dof, assign, n_dim, seed, tick, item0=0):
    """mcmc.py:163-170 with the tpCN factor of :251-279.  Returns (alpha, accept_mask)."""
    n = logl.size
    if kernel == "tpcn":
        nu = dof[assign]
        B = -0.5 * (n_dim + nu) * np.log(1 + maha_u / nu)
        A = -0.5 * (n_dim + nu) * np.log(1 + maha_up / nu)
        factor = -A + B
    else:
        factor = np.zeros(n)
    alpha = ps.metropolis_alpha(beta, logl, logl_prime, factor)
    items = np.arange(n, dtype=np.uint64) + np.uint64(item0)
    U = px.uniform1(seed, items, tick, px.TAG_ACCEPT)
    return alpha, U < alpha


def adapt(kernel, alpha, mask, assign, K, sigmas, iteration, n_dim, n_steps, n_max):
    """mcmc.py:180-194: per-cluster sigma update, then the adaptive stopping rule.
    `iteration` is the 1-based index of the step just taken.  Returns (sigmas, done, acc, target)."""
    sigma_0 = 2.38 / np.sqrt(n_dim)
    sigmas = sigmas.copy()
    for c in range(K):
        mc = assign == c
        if not mc.any():
            continue
        ma = alpha[mc].mean()
        if kernel == "tpcn":
            sigmas[c] = ps.adapt_sigma_tpcn(sigmas[c], ma, iteration, sigma_0)
        else:
            sigmas[c] = ps.adapt_sigma_rwm(sigmas[c], ma, iteration)
    acc = mask.mean()
    target = ps.adaptive_steps(n_steps, n_max, n_dim, sigmas, assign, K, acc)
    return sigmas, iteration >= target, acc, target


def prior_draw(n, n_dim, seed, tick, item0=0):
    """steps/mutate.py:102."""
    return px.uniforms(seed, np.arange(n, dtype=np.uint64) + np.uint64(item0), n_dim, tick, px.TAG_PRIOR)


def inf_repair(u, x, logl, seed, tick, item0=0):
    """steps/mutate.py:122-148 on the shared stream.  Returns (u, x, logl, n_finite)."""
    n = logl.size
    U = px.uniform1(seed, np.arange(n, dtype=np.uint64) + np.uint64(item0), tick, px.TAG_REPAIR)
    src, _ = ps.inf_repair(logl, U)
    return u[src], x[src], logl[src], int(np.sum(~np.isinf(logl)))
