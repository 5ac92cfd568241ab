"""Per-mode proposal statistics (reference: tempest/modes.py).  Means, covariances, their Cholesky
factors and inverses live on the device for the mutation kernels; the NumPy attributes of the
reference (`means`, `covariances`, `degrees_of_freedom`, `inv_covariances`, `chol_covariances`) are
lazy host copies."""
import numpy as np


class ModeStatistics:
    DOF_FALLBACK = 1e6

    def __init__(self, means, covariances, degrees_of_freedom, device=None, _dev=None):
        """From host arrays (modes.py:58-119): Cholesky + inverse per mode are computed by the HIP
        kernel with the reference's ridge-on-failure rule."""
        import torch
        if _dev is not None:       # internal: already-fitted device tensors (ctx, means, covs, chol, inv, dof[, winv])
            self._ctx, self.means_dev, self.covs_dev, self.chol_dev, self.inv_dev, self.dof_dev = _dev[:6]
            self.winv_dev = _dev[6] if len(_dev) > 6 else None      # L^-1 per mode: what the proposal kernels consume
            self._host = {}
            return
        means = np.asarray(means, dtype=np.float64)
        covs = np.asarray(covariances, dtype=np.float64)
        dof = np.asarray(degrees_of_freedom, dtype=np.float64)
        if means.ndim == 1:
            means = means.reshape(1, -1)
        if covs.ndim == 2:
            covs = covs.reshape(1, *covs.shape)
        if dof.ndim == 0:
            dof = np.array([dof])
        K, n_dim = means.shape
        if covs.shape != (K, n_dim, n_dim):
            raise ValueError(f"Covariances shape {covs.shape} incompatible with means shape {means.shape}")
        if dof.shape != (K,):
            raise ValueError(f"Degrees of freedom shape {dof.shape} incompatible with K={K}")
        from .tools import _ctx
        ctx = _ctx(n_dim) if device is None else device
        self._ctx = ctx
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)  # noqa: E731
        self.means_dev, self.covs_dev, self.dof_dev = to(means), to(covs.copy()), to(dof)
        self.chol_dev, self.inv_dev, self.winv_dev = ctx.chol_inv(self.covs_dev)
        self._host = {}

    # ------------------------------------------------------------------ reference attributes
    def _h(self, name, t):
        if name not in self._host:
            self._host[name] = t.cpu().numpy()
        return self._host[name]

    @property
    def means(self):
        return self._h("means", self.means_dev)

    @property
    def covariances(self):
        return self._h("covs", self.covs_dev)

    @property
    def degrees_of_freedom(self):
        return self._h("dof", self.dof_dev)

    @property
    def inv_covariances(self):
        return self._h("inv", self.inv_dev)

    @property
    def chol_covariances(self):
        return self._h("chol", self.chol_dev)

    @property
    def K(self) -> int:
        return int(self.means_dev.shape[0])

    @property
    def n_dim(self) -> int:
        return int(self.means_dev.shape[1])

    # ------------------------------------------------------------------------------- fitting
    @classmethod
    def _fit(cls, ctx, w_dev, n, labels_dev, K, seed, tick, dof_fallback, resample_factor, kept_count=None, comm=None,
             student_em=False):
        """Shared device path: multinomial x`resample_factor` up-sampling as multiplicities
        (modes.py:196-201 / 269-274), then median + covariance + chol/inv (student.py effective form).
        With a communicator (`comm` active: w_dev, labels_dev cover this rank's shard of the history) the label sizes,
        the draws and the fit are global."""
        import torch
        sharded = comm is not None and comm.active
        if K == 1:
            cdf = ctx.cdf_global(w_dev) if sharded else ctx.cdf(w_dev)
            draw = ctx.multinomial_counts_global if sharded else ctx.multinomial_counts
            counts = draw(cdf, seed, tick, kept_count=kept_count, factor=resample_factor,
                          n_draw_max=resample_factor * (n if not sharded else n * comm.world_size))
        else:
            # per label: weights renormalised inside the label, factor * n_label draws (modes.py:185-201)
            counts = torch.zeros(n, dtype=torch.int32, device=ctx.device)
            sizes = torch.stack([((labels_dev == k) & (w_dev > 0)).sum() for k in range(K)]).to(torch.float64)
            if sharded:
                comm.all_reduce_sum(sizes)              # rows of each label that survived trimming, over all shards
            sizes = [int(v) for v in sizes.cpu().tolist()]
            for k in range(K):
                nk = sizes[k]
                if nk == 0:
                    continue
                wk = torch.where(labels_dev == k, w_dev, torch.zeros_like(w_dev))      # masking: data movement only
                if sharded:
                    cdf = ctx.cdf_global(wk)
                    counts += ctx.multinomial_counts_global(cdf, seed, tick + k, kept_count=None, factor=resample_factor,
                                                            n_draw_max=resample_factor * nk)
                else:
                    cdf = ctx.cdf(wk)
                    counts += ctx.multinomial_counts(cdf, seed, tick + k, kept_count=None, factor=resample_factor,
                                                     n_draw_max=resample_factor * nk)
        means, covs, chol, inv, winv = ctx.fit_modes(counts, labels_dev, K, n, global_=sharded)
        dof = torch.full((K,), float(dof_fallback), dtype=torch.float64, device=ctx.device)   # nu = inf -> fallback (F5)
        ms = cls(None, None, None, _dev=(ctx, means, covs, chol, inv, dof, winv))
        return ms._student_em(counts, labels_dev, n, dof_fallback, comm=comm) if student_em else ms

    def _student_em(self, counts, labels_dev, n, dof_fallback, comm=None):
        """Opt-in extension (Sampler(student_em=True)): every mode's (mu, Sigma, nu) refined by the Student-t EM of
        tempest/student.py:66-116 with a working degrees-of-freedom update, from the default estimator as the start, over the
        up-sampled rows (counts = multiplicities) of the history on the device; nu = inf keeps dof_fallback."""
        import torch
        from .device import KEY_U
        from .student import em_on_device
        if comm is not None and comm.active:
            raise NotImplementedError("student_em=True is not available on a sharded run")
        ctx = self._ctx
        ptr, ld = ctx.history_ptr(KEY_U)
        means, covs = self.means_dev.cpu().numpy().copy(), self.covs_dev.cpu().numpy().copy()
        dof = np.full(self.K, float(dof_fallback))
        for k in range(self.K):
            mu, Sigma, nu, _ = em_on_device(ctx, ptr, ld, n, counts, labels_dev if self.K > 1 else None, k, means[k], covs[k])
            means[k], covs[k] = mu, Sigma
            if np.isfinite(nu):
                dof[k] = nu
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)  # noqa: E731
        covs_dev = to(covs)
        chol, inv, winv = ctx.chol_inv(covs_dev)
        return ModeStatistics(None, None, None, _dev=(ctx, to(means), covs_dev, chol, inv, to(dof), winv))

    @classmethod
    def from_particles(cls, u, weights, labels, dof_fallback: float = DOF_FALLBACK, resample_factor: int = 4,
                       seed=None):
        """Per-cluster fit from host arrays (modes.py:131-219)."""
        import torch
        u = np.asarray(u, dtype=np.float64)
        weights = np.asarray(weights, dtype=np.float64)
        labels = np.asarray(labels)
        if u.shape[0] != weights.shape[0] or u.shape[0] != labels.shape[0]:
            raise ValueError("u, weights, and labels must have compatible shapes")
        uniq, dense = np.unique(labels, return_inverse=True)
        from .tools import _ctx
        ctx = _ctx(u.shape[1])
        n = u.shape[0]
        ctx.history_load(u, None, np.zeros(n), [0.0], [0.0], [n])
        w = torch.from_numpy(weights / weights.sum()).to(ctx.device)
        lab = torch.from_numpy(dense.astype(np.int32)).to(ctx.device)
        seed = int(np.random.randint(0, 2 ** 62)) if seed is None else seed
        return cls._fit(ctx, w, n, lab if uniq.size > 1 else None, int(uniq.size), seed, 0, dof_fallback,
                        resample_factor)

    @classmethod
    def from_global(cls, u, weights, dof_fallback: float = DOF_FALLBACK, resample_factor: int = 4, seed=None):
        """Single global mode from host arrays (modes.py:221-288)."""
        import torch
        u = np.asarray(u, dtype=np.float64)
        weights = np.asarray(weights, dtype=np.float64)
        if u.shape[0] != weights.shape[0]:
            raise ValueError("u and weights must have same length")
        from .tools import _ctx
        ctx = _ctx(u.shape[1])
        n = u.shape[0]
        ctx.history_load(u, None, np.zeros(n), [0.0], [0.0], [n])
        w = torch.from_numpy(weights / weights.sum()).to(ctx.device)
        seed = int(np.random.randint(0, 2 ** 62)) if seed is None else seed
        return cls._fit(ctx, w, n, None, 1, seed, 0, dof_fallback, resample_factor)

    def __repr__(self) -> str:
        return f"ModeStatistics(K={self.K}, n_dim={self.n_dim}, dof={self.degrees_of_freedom})"
