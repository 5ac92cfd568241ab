"""Reweighting step: choose the next inverse temperature and weight the whole history
(reference: tempest/steps/reweight.py).

Each trial beta costs one streaming reduction over (logl, cached log-mixture) on the device
(`StateManager.reweight_eval`), returning (max, s1, s2) from which ESS = s1^2/s2 and
logZ = max + log s1.  The bracket search and the bisection are the reference's host logic,
decision for decision (reweight.py:123-297); only the weights of the chosen beta are materialised.
"""
import math
from typing import Optional

import numpy as np


class DeviceWeights:
    """Normalised importance weights living on the GPU.  Behaves like a NumPy array where host code
    needs one (len, np.sum, np.asarray), while the next steps read `.dev` without any copy."""

    def __init__(self, dev, beta=None, triple=None):
        self.dev = dev
        self.beta = beta
        self.triple = triple

    def __len__(self):
        return int(self.dev.numel())

    @property
    def shape(self):
        return (len(self),)

    def __array__(self, dtype=None, copy=None):
        a = self.dev.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a


class Reweighter:
    """Same constructor and `run()` contract as the reference's Reweighter (reweight.py:53-86,341-495)."""

    _MAX_BISECTION_ITERATIONS = 200

    def __init__(self, state, pbar=None, n_particles: int = 256, ess_ratio: float = 2.0,
                 volume_variation: Optional[float] = None, ESS_TOLERANCE: float = 0.001,
                 BETA_TOLERANCE: float = 1e-5, BETA_RTOL: float = 1e-8, METRIC_ATOL: float = 0.5,
                 METRIC_ATOL_CV: float = 0.01):
        self.state = state
        self.pbar = pbar
        self.n_particles = n_particles
        self.ess_ratio = ess_ratio
        self.volume_variation = volume_variation
        self.target_metric = volume_variation if volume_variation is not None else ess_ratio * n_particles
        self.ESS_TOLERANCE = ESS_TOLERANCE
        self.BETA_TOLERANCE = BETA_TOLERANCE
        self.BETA_RTOL = BETA_RTOL
        self.METRIC_ATOL = METRIC_ATOL
        self.METRIC_ATOL_CV = METRIC_ATOL_CV
        self.n_evals = 0          # trial betas evaluated
        self.n_passes = 0         # passes over the history (a pass evaluates up to 16 betas)
        self.batch_depth = None   # bisection levels evaluated per pass: None = automatic, 1 = one beta per pass (reference)
        self._cache = {}

    # ------------------------------------------------------------------ trial betas
    def _eval_many(self, betas):
        """Evaluate the not-yet-known betas of `betas` in ONE pass over the history (<= 16 per pass)."""
        todo = []
        for b in betas:
            if b not in self._cache and b not in todo:
                todo.append(b)
        for i in range(0, len(todo), 16):
            chunk = todo[i:i + 16]
            res = self.state.reweight_eval(chunk)
            self.n_passes += 1
            for b, (m, s1, s2) in zip(chunk, res):
                self._cache[b] = (float(m), float(s1), float(s2), float(s1 * s1 / s2))
                self.n_evals += 1

    def _eval(self, beta: float):
        """(vmax, s1, s2, ess) at beta, memoised within one run() (the reference re-evaluates the final
        beta several times; the result is identical)."""
        hit = self._cache.get(beta)
        if hit is None:
            self._eval_many([beta])
            hit = self._cache[beta]
        return hit

    @staticmethod
    def _midpoint_tree(lo: float, hi: float, depth: int):
        """All midpoints a bisection started on [lo, hi] can visit in its next `depth` levels, computed with the
        reference's own expression (hi + lo) * 0.5 on the exact bracket ends, so they are bit-identical to the
        sequential ones whatever the decisions turn out to be (exact-bisection batching)."""
        out, level = [], [(lo, hi)]
        for _ in range(depth):
            nxt = []
            for a, b in level:
                mid = (b + a) * 0.5
                out.append(mid)
                nxt.append((a, mid))
                nxt.append((mid, b))
            level = nxt
        return out

    def _depth(self) -> int:
        """Levels per pass.  A pass costs a fixed latency (launch + sync + 24-byte copy, ~90 us) plus ~2.2 ns per
        history row and trial beta (one FP64 exp each; measured on MI355X), and buys `depth` bisection levels for
        2^depth - 1 betas: pick the depth with the lowest cost per level."""
        if self.batch_depth is not None:
            return self.batch_depth
        rows = max(1, self.state.ctx.size)
        best, best_cost = 1, None
        for k in (1, 2, 3, 4):
            cost = (90.0 + max(rows * 4e-6, rows * 2.2e-6 * (2 ** k - 1))) / k
            if best_cost is None or cost < best_cost:
                best, best_cost = k, cost
        return best

    def _prefetch(self, lo: float, hi: float, extra=()):
        """Before the sequential logic asks for the midpoint of [lo, hi]: evaluate the coming `batch_depth`
        levels of candidates in one pass (15 betas for depth 4) unless that midpoint is already known."""
        mid = (hi + lo) * 0.5
        bd = self._depth()
        if bd > 1 and mid not in self._cache:
            depth = bd if not extra else min(bd, 3)
            self._eval_many(list(extra) + self._midpoint_tree(lo, hi, depth))

    def _weights_dev(self, beta: float):
        m, s1, _, _ = self._eval(beta)
        ctx = self.state.ctx
        if self.state.comm is not None and self.state.comm.active:
            pass   # m, s1 are already global: local weights are normalised by the global sum
        return ctx.weights(beta, m, s1)

    def _cv(self, beta: float) -> float:
        from ..tools import device_volume_variation
        return device_volume_variation(self.state.ctx, self._weights_dev(beta), self.state.n_history_global(),
                                       self.state.comm)

    def _compute_metric_and_weights(self, beta: float) -> tuple:
        """(weights, ess, metric) like reweight.py:88-118; weights stay on the device (normalised)."""
        ess = self._eval(beta)[3]
        if self.volume_variation is not None:
            metric = self._cv(beta)
        else:
            metric = ess
        return DeviceWeights(self._weights_dev(beta), beta), ess, metric

    # --------------------------------------------------------------------- searches
    def _tol(self, lo, hi):
        scale = max(abs(lo), abs(hi), np.finfo(float).tiny)
        return max(self.BETA_RTOL * scale, self.BETA_TOLERANCE * scale)

    def _find_beta_bisection(self, beta_min: float, beta_max: float, target: float, metric_fn) -> tuple:
        """reweight.py:123-223."""
        dynamic = self.volume_variation is not None
        beta, aux = None, None
        for _ in range(self._MAX_BISECTION_ITERATIONS):
            beta = (beta_max + beta_min) * 0.5
            if not dynamic:
                self._prefetch(beta_min, beta_max)
            metric_val, aux = metric_fn(beta)
            if not np.isfinite(metric_val):
                metric_val = 1e10
            atol = self.METRIC_ATOL_CV if dynamic else self.METRIC_ATOL
            metric_converged = abs(metric_val - target) < max(self.ESS_TOLERANCE * abs(target), atol)
            beta_converged = (beta_max - beta_min) < self._tol(beta_min, beta_max)
            if metric_converged or beta_converged or beta == 1.0:
                return beta, aux
            if not dynamic:
                if metric_val < target:      # ESS falls as beta grows
                    beta_max = beta
                else:
                    beta_min = beta
            else:
                if metric_val < target:      # CV grows with beta
                    beta_min = beta
                else:
                    beta_max = beta
        return beta, aux

    def _find_ess_bracket(self, beta_current: float, ess_target: float) -> tuple:
        """reweight.py:225-297: (low, high) with ESS(low) >= target > ESS(high); equal when no crossing."""
        beta_low, beta_high = beta_current, 1.0
        if self._depth() > 1:      # both ends and the first levels of the bracket search in one pass
            self._prefetch(beta_low, beta_high, extra=(beta_current, 1.0))
        if self._eval(beta_current)[3] <= ess_target:
            return beta_current, beta_current
        if self._eval(1.0)[3] >= ess_target:
            return 1.0, 1.0
        while True:
            beta_mid = (beta_high + beta_low) * 0.5
            if (beta_high - beta_low) <= self._tol(beta_low, beta_high):
                break
            self._prefetch(beta_low, beta_high)
            if self._eval(beta_mid)[3] >= ess_target:
                beta_low = beta_mid
            else:
                beta_high = beta_mid
        return beta_low, beta_high

    # -------------------------------------------------------------------------- run
    def run(self):
        st = self.state
        self._cache = {}
        it = st.get_current("iter") + 1
        st.set_current("iter", it)
        if self.pbar is not None:
            self.pbar.update_iter()

        if st.get_history_length() == 0:      # first iteration contract (reweight.py:365-383)
            st.update_current({"beta": 0.0, "logz": 0.0, "ess": self.ess_ratio * self.n_particles, "cv": 0.0})
            if self.pbar is not None:
                self.pbar.update_stats(dict(beta=0.0, ESS=int(self.ess_ratio * self.n_particles), logZ=0.0, CV=0.0))
            # uniform weights 1/N (reweight.py:383) as a read-only broadcast view: materialising two million-entry host arrays
            # cost 80 ms of the first iteration (first-touch page faults) for values nobody reads at beta = 0
            return np.broadcast_to(np.float64(1.0 / self.n_particles), (self.n_particles,))

        beta_prev = st.get_current("beta")
        ess_target = self.ess_ratio * self.n_particles
        beta_low, beta_high = self._find_ess_bracket(beta_prev, ess_target)

        if self.volume_variation is None:
            if beta_low == beta_high:
                beta = beta_low
            else:
                beta, _ = self._find_beta_bisection(beta_prev, beta_high, ess_target,
                                                    lambda b: (self._eval(b)[3], None))
        else:
            if beta_low == beta_high:
                beta = beta_low
            else:
                vv_prev, vv_high = self._cv(beta_prev), self._cv(beta_high)
                if self.volume_variation >= vv_high:
                    beta = beta_high
                elif self.volume_variation <= vv_prev:
                    beta = beta_prev
                else:
                    beta, _ = self._find_beta_bisection(beta_prev, beta_high, self.volume_variation,
                                                        lambda b: (self._cv(b), None))
        m, s1, _, ess = self._eval(beta)
        w = self._weights_dev(beta)
        from ..tools import device_volume_variation
        cv = device_volume_variation(st.ctx, w, st.n_history_global(), st.comm)
        logz = m + math.log(s1)
        if self.pbar is not None:
            self.pbar.update_stats(dict(beta=beta, ESS=int(ess), logZ=logz, CV=cv))
        st.update_current({"logz": logz, "beta": beta, "ess": ess, "cv": cv})
        return DeviceWeights(w, beta, (m, s1))
