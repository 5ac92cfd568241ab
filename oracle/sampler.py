"""Vectorised NumPy Persistent Sampling sampler: the oracle's end-to-end loop.

TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` "port" leg of bench.py).  Restates
tempest/core.py:110-185,360-374 + the four steps (tempest/steps/*.py) with clustering=False, using the
pure functions of oracle/ps.py and oracle/mcmc.py.  The reference's per-walker Python loops are
vectorised over walkers, and its global NumPy RNG is replaced by the shared Philox stream consumed in
the same order as the device path (one tick per RNG-consuming launch), so a device run and an oracle
run with the same seed can be compared iteration by iteration.

`compute_logw_and_logz` is restated as the reference writes it (the N_h x T matrix is rebuilt on every
evaluation, state_manager.py:466): that is the cost structure the CPU baseline reports.
"""
import time

import numpy as np

from . import mcmc as omc
from . import philox as px
from . import ps


class OracleSampler:
    def __init__(self, prior_transform, log_likelihood, n_dim, n_particles, ess_ratio=2.0, sample="tpcn",
                 resample="mult", n_steps=1, n_max_steps=None, periodic=None, reflective=None, seed=0):
        self.prior, self.loglike = prior_transform, log_likelihood
        self.d, self.n = n_dim, n_particles
        self.ess_ratio, self.kernel, self.resample = ess_ratio, sample, resample
        self.n_steps = n_steps
        self.n_max = 20 * n_steps if n_max_steps is None else n_max_steps
        self.flags = omc.bc_flags(n_dim, periodic, reflective)
        self.seed, self.tick = int(seed), 0
        self.hist = dict(u=[], x=[], logl=[], beta=[], logz=[], steps=[], ess=[], acceptance=[], efficiency=[])
        self.cur = dict(iter=0, calls=0, beta=0.0, logz=0.0)
        self.timing = dict(reweight=0.0, train=0.0, resample=0.0, mutate=0.0)
        self.pms = 0

    def _tick(self):
        self.tick += 1
        return self.tick

    # ------------------------------------------------------------------ history helpers
    def _flat(self):
        h = self.hist
        return (np.concatenate(h["u"]), np.concatenate(h["x"]), np.concatenate(h["logl"]), np.array(h["beta"]),
                np.array(h["logz"]), np.array([len(a) for a in h["logl"]]))

    def _ess_logz(self, beta):
        _, _, logl, bt, zt, nt = self._flat()
        logw, logz = ps.compute_logw_and_logz(logl, bt, zt, nt, beta)
        w = np.exp(logw - np.max(logw))
        return ps.effective_sample_size(w), logz, w

    # ------------------------------------------------------------------------ one iteration
    def sample(self):
        cur, h = self.cur, self.hist
        cur["iter"] += 1
        t0 = time.perf_counter()
        # reweight (steps/reweight.py:341-426, ESS mode; the cv diagnostic is skipped)
        if len(h["beta"]) == 0:
            beta, w, ess, logz = 0.0, None, self.ess_ratio * self.n, 0.0
        else:
            target = self.ess_ratio * self.n
            ess_fn = lambda b: self._ess_logz(b)[0]      # noqa: E731
            lo, hi = ps.find_ess_bracket(ess_fn, cur["beta"], target)
            if lo == hi:
                beta = lo
            else:
                beta, _ = ps.find_beta_bisection(lambda b: (ess_fn(b), None), cur["beta"], hi, target)
            ess, logz, w = self._ess_logz(beta)
            w = w / w.sum()
        cur.update(beta=beta, ess=ess, logz=logz)
        t1 = time.perf_counter()
        self.timing["reweight"] += t1 - t0
        if beta == 0.0:
            # steps/mutate.py:99-149
            u = omc.prior_draw(self.n, self.d, self.seed, self._tick())
            x = self.prior(u)
            logl = np.asarray(self.loglike(x), dtype=np.float64)
            cur["calls"] += self.n
            n_inf_before = np.isinf(logl).sum()
            u, x, logl, nfin = omc.inf_repair(u, x, logl, self.seed, self._tick())
            if n_inf_before:
                with np.errstate(divide="ignore"):
                    cur["logz"] += float(np.log(nfin / self.n))
            steps, acc, eff = 1, 1.0, 1.0
            self.timing["mutate"] += time.perf_counter() - t1
        else:
            uh, xh, lh, _, _, _ = self._flat()
            # train (steps/train.py:91-122 without clustering; modes.py:221-288; student.py effective form)
            thr, ksum, kcnt = ps.trim_threshold_sorted(w, ps.TRIM_ESS, ps.TRIM_BINS, normalized=True)
            wt = np.where(w >= thr, w, 0.0)
            U = px.uniform1(self.seed, np.arange(4 * kcnt, dtype=np.uint64), self._tick(), px.TAG_UPSAMPLE)
            counts = np.bincount(ps.multinomial_resample(wt, U), minlength=w.size)
            mu, Sig, _ = ps.median_cov_from_counts(uh, counts)
            _, chol, inv = ps.mode_statistics(mu[None], Sig[None])
            means, dof = mu[None], np.array([ps.DOF_FALLBACK])
            t2 = time.perf_counter()
            self.timing["train"] += t2 - t1
            # resample (steps/resample.py:78-99)
            if self.resample == "mult":
                U = px.uniform1(self.seed, np.arange(self.n, dtype=np.uint64), self._tick(), px.TAG_RESAMPLE)
                idx = ps.multinomial_resample(w, U)
            else:
                u0 = px.uniform1(self.seed, np.zeros(1, dtype=np.uint64), self._tick(), px.TAG_SYST)[0]
                idx = ps.systematic_resample(self.n, w, u0)
            u, x, logl = uh[idx].copy(), xh[idx].copy(), lh[idx].copy()
            t3 = time.perf_counter()
            self.timing["resample"] += t3 - t2
            # mutate (mcmc.py:142-208)
            assign = np.zeros(self.n, dtype=np.int64)
            sigma_0 = 2.38 / np.sqrt(self.d)
            sigmas = np.array([min(sigma_0, 0.99) if self.kernel == "tpcn" else sigma_0])
            it = 0
            while True:
                it += 1
                up, mu_, mup = omc.propose(self.kernel, u, assign, means, chol, inv, dof, sigmas, self.flags,
                                           self.seed, self._tick())
                xp = self.prior(up)
                lp = np.asarray(self.loglike(xp), dtype=np.float64)
                cur["calls"] += self.n
                alpha, mask = omc.accept(self.kernel, beta, logl, lp, mu_, mup, dof, assign, self.d, self.seed,
                                         self._tick())
                u[mask], x[mask], logl[mask] = up[mask], xp[mask], lp[mask]
                sigmas, done, _, _ = omc.adapt(self.kernel, alpha, mask, assign, 1, sigmas, it, self.d, self.n_steps,
                                               self.n_max)
                if done:
                    self._tick()          # the device had already launched the next proposal (speculatively) when it
                    break                 # read the stop flag: that launch's tick is spent
            steps, acc, eff = it, float(alpha.mean()), float(sigmas.mean() / sigma_0)
            self.pms += it * self.n
            self.timing["mutate"] += time.perf_counter() - t3
        cur.update(steps=steps, acceptance=acc, efficiency=eff)
        for k, v in (("u", u), ("x", x), ("logl", logl)):
            h[k].append(v)
        for k in ("beta", "logz", "steps", "ess", "acceptance", "efficiency"):
            h[k].append(cur[k])
        return cur

    def not_terminated(self, n_total):
        if not self.hist["beta"]:
            return True
        ess, _, _ = self._ess_logz(1.0)
        return 1.0 - self.cur["beta"] >= 1e-4 or ess < n_total

    def run(self, n_total=4096, max_iter=None):
        k = 0
        while self.not_terminated(n_total) and (max_iter is None or k < max_iter):
            self.sample()
            k += 1
        if self.hist["beta"]:
            self.cur["logz"] = self._ess_logz(1.0)[1]
        return self.cur["logz"]

    def posterior_moments(self):
        _, xh, _, _, _, _ = self._flat()
        _, _, w = self._ess_logz(1.0)
        w = w / w.sum()
        m = np.average(xh, weights=w, axis=0)
        return m, np.average((xh - m) ** 2, weights=w, axis=0)
