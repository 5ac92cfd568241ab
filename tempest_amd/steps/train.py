"""Training step: fit the proposal statistics for the mutation kernels from the weighted history
(reference: tempest/steps/train.py:65-127; tempest/modes.py; tempest/student.py -- effective form).

Everything after the weights stays on the device and needs no host synchronisation:
trim threshold (sort + prefix sums, tools.py:10-55) -> masked prefix sum -> x4 multinomial up-sampling as
multiplicities (modes.py:196-201) -> per-dimension median, covariance, Cholesky and inverse.
"""
import numpy as np

from ..modes import ModeStatistics
from .resample import _as_device_weights


class Trainer:
    def __init__(self, state, pbar=None, clusterer=None, cluster_every: int = 1, clustering: bool = True,
                 TRIM_ESS: float = 512, TRIM_BINS: int = 10, DOF_FALLBACK: float = 1.0, rng=None, student_em: bool = False):
        self.state = state
        self.pbar = pbar
        self.clusterer = clusterer
        self.cluster_every = cluster_every
        self.clustering = clustering
        self.TRIM_ESS = TRIM_ESS
        self.TRIM_BINS = TRIM_BINS
        self.DOF_FALLBACK = DOF_FALLBACK
        self.student_em = bool(student_em)      # opt-in extension: working Student-t EM per mode (tempest_amd/student.py)
        self.rng = rng

    def _rng(self):
        if self.rng is None:
            from ..mcmc import PhiloxStream
            self.rng = PhiloxStream(np.random.randint(0, 2 ** 62))
        return self.rng

    def run(self, weights) -> ModeStatistics:
        import torch
        st = self.state
        n_dim = st.n_dim
        if st.get_current("beta") == 0.0:    # dummy statistics, unused at beta = 0 (train.py:80-88)
            return ModeStatistics(np.zeros((1, n_dim)), np.eye(n_dim).reshape(1, n_dim, n_dim),
                                  np.array([self.DOF_FALLBACK]), device=st.ctx)
        ctx = st.ctx
        ctx.use_current_stream()
        rng = self._rng()
        w = _as_device_weights(weights, ctx)
        n_h = w.numel()
        # Every call below is the *_global form: on one GPU it is the plain kernel; with a communicator attached the
        # threshold, the up-sampling draws and the fit are those of the WHOLE weighted history (train.py:91-122), so a
        # sharded run trains the same proposal as the one-GPU run.
        thr = ctx.trim_threshold(w, self.TRIM_ESS, self.TRIM_BINS, global_=True)   # (threshold, kept_sum, kept_count, ess)
        n_draw_max = 4 * st.n_history_global()
        K, labels = 1, None
        if self.clustering and self.clusterer is not None:
            it = st.get_current("iter")
            refit = (it % self.cluster_every == 0) or it == 0
            labels, K = self.clusterer.fit_predict_device(st, w, thr, refit, rng)
        if K > 1:
            wt = torch.where(w >= thr[0], w, torch.zeros_like(w))       # trimmed weights, history order
            tick = rng.next()
            for _ in range(K - 1):
                rng.next()
            ms = ModeStatistics._fit(ctx, wt, n_h, labels, K, rng.seed, tick, self.DOF_FALLBACK, 4, comm=st.comm, student_em=self.student_em)
        else:
            cdf = ctx.cdf_global(w, thr[0:1])
            counts = ctx.multinomial_counts_global(cdf, rng.seed, rng.next(), kept_count=thr[2:3], factor=4,
                                                   n_draw_max=n_draw_max)
            means, covs, chol, inv, winv = ctx.fit_modes(counts, None, 1, n_h, global_=True)
            dof = torch.full((1,), float(self.DOF_FALLBACK), dtype=torch.float64, device=ctx.device)
            ms = ModeStatistics(None, None, None, _dev=(ctx, means, covs, chol, inv, dof, winv))
            if self.student_em:
                ms = ms._student_em(counts, None, n_h, self.DOF_FALLBACK, comm=st.comm)
        if self.pbar is not None:
            self.pbar.update_stats(dict(K=ms.K))
        return ms
