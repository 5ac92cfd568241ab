"""Run coordinator (reference: tempest/core.py): wires the four steps around one StateManager, runs the
Persistent Sampling loop, evaluates the stopping rule and assembles posterior()/evidence().

The user's callbacks are adapted once (`CallbackAdapter`) to the device layout: particle tensors are
(n_dim, n) SoA in HBM and a callback sees `tensor.T`, an (n, n_dim) strided view, so `x[:, ::2]` etc. work
without any copy.  backend="torch" hands torch-ROCm tensors to the callbacks (no host round trip),
backend="numpy" stages through the host for NumPy likelihoods (drop-in, PCIe-bound), "auto" probes.
"""
import warnings
from pathlib import Path
from typing import Optional, Union

import numpy as np

from .config import (BETA_RTOL, BETA_TOLERANCE, DOF_FALLBACK, ESS_TOLERANCE, METRIC_ATOL, METRIC_ATOL_CV, TRIM_BINS,
                     TRIM_ESS, SamplerConfig)
from .mcmc import PhiloxStream
from .state_manager import StateManager


class CallbackAdapter:
    """prior_transform / log_likelihood of the reference's contract -> device callables on SoA tensors."""

    def __init__(self, config: SamplerConfig, device, get_distribute_func):
        self.cfg = config
        self.device = device
        self.backend = None if config.backend == "auto" else config.backend
        self.batch_prior = config.batch_prior
        self._distribute = get_distribute_func
        from .hipcallbacks import fused_plugin
        self.hip_plugin = fused_plugin(config.prior_transform, config.log_likelihood)   # HipCallbacks pair -> fused step
        for cb in (config.prior_transform, config.log_likelihood):                       # host inputs land on OUR device
            owner = getattr(cb, "__self__", None)
            if owner is not None and hasattr(owner, "_soa"):
                owner.device = device

    # ------------------------------------------------------------------ probing
    def _probe(self):
        import torch
        cfg = self.cfg
        d = cfg.n_dim
        us = torch.linspace(0.05, 0.95, 4 * d, dtype=torch.float64, device=self.device).reshape(d, 4).T  # (4, d) view
        if self.backend is None:
            self.backend = "torch"
            try:
                xs = cfg.prior_transform(us)
                ok = isinstance(xs, torch.Tensor) and tuple(xs.shape) == (4, d) and xs.device == us.device
                if not ok:
                    xs0 = cfg.prior_transform(us[0])
                    ok = isinstance(xs0, torch.Tensor) and tuple(xs0.shape) == (d,)
                    xs = torch.stack([cfg.prior_transform(r) for r in us]) if ok else None
                if ok and cfg.vectorize:
                    ll = cfg.log_likelihood(xs)
                    ok = isinstance(ll, torch.Tensor) and ll.numel() == 4
                elif ok:
                    ok = False          # per-sample likelihoods run on the host
            except Exception:
                ok = False
            if not ok:
                self.backend = "numpy"
        if self.batch_prior is None:
            try:
                if self.backend == "torch":
                    xb = cfg.prior_transform(us)
                    xr = torch.stack([cfg.prior_transform(r) for r in us])
                    self.batch_prior = bool(isinstance(xb, torch.Tensor) and xb.shape == xr.shape
                                            and torch.equal(xb, xr))
                else:
                    uh = us.cpu().numpy().copy()
                    xb = np.asarray(cfg.prior_transform(uh))
                    xr = np.array([cfg.prior_transform(r) for r in uh])
                    self.batch_prior = bool(xb.shape == xr.shape and np.array_equal(xb, xr))
            except Exception:
                self.batch_prior = False
            if not self.batch_prior:
                warnings.warn("prior_transform is not batch-safe: it will be applied row by row "
                              "(the reference's convention), which serialises every MCMC step.", stacklevel=3)

    # ---------------------------------------------------------------- callbacks
    def prior(self, u_soa):
        """(d, n) SoA unit-cube tensor -> (d, n) SoA parameter tensor."""
        import torch
        if self.backend is None or self.batch_prior is None:
            self._probe()
        if self.backend == "torch":
            uv = u_soa.T
            xv = self.cfg.prior_transform(uv) if self.batch_prior else torch.stack(
                [self.cfg.prior_transform(r) for r in uv])
            if xv.dtype != torch.float64:
                xv = xv.to(torch.float64)
            xt = xv.T
            return xt if xt.is_contiguous() else xt.contiguous()
        uh = np.ascontiguousarray(u_soa.cpu().numpy().T)
        xh = np.asarray(self.cfg.prior_transform(uh)) if self.batch_prior else np.array(
            [self.cfg.prior_transform(r) for r in uh])
        return torch.from_numpy(np.ascontiguousarray(np.asarray(xh, dtype=np.float64).T)).to(self.device)

    def loglike(self, x_soa, return_blobs=False):
        """(d, n) SoA parameter tensor -> (n,) log-likelihood tensor (FP64, on the device); with `return_blobs` the pair
        (tensor, blobs): the likelihood's auxiliary outputs as a host array (None without them), core.py:317-358."""
        import torch
        if self.backend is None:
            self._probe()
        cfg = self.cfg
        if cfg.vectorize and self.backend == "torch":
            ll = cfg.log_likelihood(x_soa.T)
            if ll.dtype != torch.float64:
                ll = ll.to(torch.float64)
            ll = ll.reshape(-1).contiguous()
            return (ll, None) if return_blobs else ll
        xh = np.ascontiguousarray(x_soa.cpu().numpy().T)
        ll, blobs = self.loglike_host(xh)
        ll = torch.from_numpy(np.ascontiguousarray(ll, dtype=np.float64).reshape(-1)).to(self.device)
        return (ll, blobs) if return_blobs else ll

    def loglike_host(self, x):
        """The reference's SamplerCore._log_like on host arrays (core.py:317-358)."""
        cfg = self.cfg
        if cfg.vectorize:
            return np.asarray(cfg.log_likelihood(x)), None
        if cfg.pool is not None:
            results = list(self._distribute()(cfg.log_likelihood, x))
        else:
            results = list(map(cfg.log_likelihood, x))
        if results and isinstance(results[0], (tuple, list)) and len(results[0]) > 1:
            logl = np.array([float(r[0]) for r in results])
            return logl, _pack_blobs([r[1:] for r in results], cfg.blobs_dtype)
        return np.array([float(v) for v in results]), None


def _pack_blobs(rows, dtype=None):
    """The per-sample auxiliary outputs of a non-vectorised likelihood as ONE host array with the sample axis first.
    The element type is the configured `blobs_dtype`, else that of the first sample (text and anything NumPy cannot type
    become `object`); axes of length one behind the sample axis are dropped, so one scalar per sample gives shape (n,).
    Behaviour of the reference's blob handling (core.py:335-352); host data the device path never sees."""
    if dtype is None:
        try:
            dtype = np.asarray(rows[0]).dtype
        except ValueError:               # ragged sample
            dtype = np.dtype(object)
        if dtype.kind in ("U", "S"):
            dtype = np.dtype(object)
    arr = np.array(rows, dtype=dtype)
    shape = arr.shape[:1] + tuple(m for m in arr.shape[1:] if m != 1)
    return arr if shape == arr.shape else arr.reshape(shape)


class SamplerCore:
    """Internal coordinator; `Sampler` delegates to it (core.py:20-108)."""

    def __init__(self, config: SamplerConfig, state: StateManager):
        from .steps import Mutator, Resampler, Reweighter, Trainer
        self.config = config
        self.state = state
        comm = state.comm
        self.world = comm.world_size if comm is not None else 1
        if config.n_particles % self.world:
            raise ValueError(f"n_particles ({config.n_particles}) must be divisible by the number of GPUs ({self.world})")
        self.n_local = config.n_particles // self.world
        # random_state seeds the counter-based stream; without it the seed is taken from NumPy's global
        # stream, so `np.random.seed(k)` makes a run reproducible the way it does for the reference
        seed = config.random_state if config.random_state is not None else int(np.random.randint(0, 2 ** 62))
        if comm is not None and comm.active:      # every rank must walk the same counter-based stream
            seed = comm.sum_int(seed if comm.rank == 0 else 0)
        self.rng = PhiloxStream(seed)
        self.callbacks = None

        self.reweighter = Reweighter(state=state, pbar=None, n_particles=config.n_particles, ess_ratio=config.ess_ratio,
                                     volume_variation=config.volume_variation, ESS_TOLERANCE=ESS_TOLERANCE,
                                     BETA_TOLERANCE=BETA_TOLERANCE, BETA_RTOL=BETA_RTOL, METRIC_ATOL=METRIC_ATOL,
                                     METRIC_ATOL_CV=METRIC_ATOL_CV)
        clusterer = None
        if config.clustering:
            from .cluster import HierarchicalGaussianMixture
            clusterer = HierarchicalGaussianMixture(
                n_init=1,
                max_iterations=1000 if config.n_max_clusters is None else config.n_max_clusters - 1,
                min_points=None if config.n_max_clusters is None else 4 * config.n_dim,
                threshold_modifier=config.split_threshold, covariance_type="full", verbose=False,
                normalize=config.normalize)
        self.trainer = Trainer(state=state, pbar=None, clusterer=clusterer, cluster_every=config.cluster_every,
                               clustering=config.clustering, TRIM_ESS=TRIM_ESS, TRIM_BINS=TRIM_BINS,
                               DOF_FALLBACK=DOF_FALLBACK, rng=self.rng, student_em=config.student_em)
        self.resampler = Resampler(state=state, n_particles=self.n_local, resample=config.resample,
                                   clusterer=clusterer, clustering=config.clustering,
                                   have_blobs=config.blobs_dtype is not None, rng=self.rng)
        self.mutator = Mutator(state=state, prior_transform=config.prior_transform, log_likelihood=self._log_like,
                               pbar=None, n_particles=self.n_local, n_dim=config.n_dim, n_steps=config.n_steps,
                               n_max_steps=config.n_max_steps, sampler=config.sample, periodic=config.periodic,
                               reflective=config.reflective, have_blobs=config.blobs_dtype is not None, rng=self.rng,
                               graph=config.graph)
        self.pbar = None
        self.t0 = 0
        self.timing = {"reweight": 0.0, "train": 0.0, "resample": 0.0, "mutate": 0.0, "commit": 0.0}

    # ---------------------------------------------------------------------------- run loop
    def _ensure_callbacks(self):
        w = getattr(self, "_warm", None)
        if w is not None:
            w.wait()                      # the process's one-time start-up, running in parallel since the construction
        if self.callbacks is None:
            self.callbacks = CallbackAdapter(self.config, self.state.device, self._get_distribute_func)
            self.mutator.device_callbacks = (self.callbacks.prior, self.callbacks.loglike)

    def run_sampling(self, n_total: int = 4096, progress: bool = True,
                     resume_state_path: Optional[Union[str, Path]] = None, save_every: Optional[int] = None) -> None:
        if resume_state_path is not None:
            self._initialize_from_resume(resume_state_path)
            it = self.state.get_current("iter")
            t0 = int(it) if it is not None else 0
            if it is None:
                self.state.set_current("iter", t0)
        else:
            t0 = 0
            self._initialize_fresh()
        self.n_total = int(n_total)
        self.t0 = t0
        from .tools import ProgressBar
        show = progress and (self.state.comm is None or self.state.comm.rank == 0)
        self.pbar = ProgressBar(show, initial=t0)
        self._update_progress_bar_initial()
        self.reweighter.pbar = self.pbar
        self.trainer.pbar = self.pbar
        self.mutator.pbar = self.pbar if show else None

        while self._not_termination():
            self.execute_iteration(save_every=save_every, t0=t0, return_state=False)

        _, logz = self._logz_at(1.0)
        self.state.set_current("logz", logz)
        self.logz_err = None
        if save_every is not None:
            self.save_sampler_state(self._checkpoint_path("final"))
        self.pbar.close()

    def _checkpoint_path(self, tag: str) -> Path:
        """{output_dir}/{label}_{tag}.state (core.py:154-171); a sharded run writes a checkpoint directory, `.ckpt`."""
        comm = self.state.comm
        sharded = comm is not None and comm.active
        return self.config.output_dir / f"{self.config.output_label}_{tag}{'.ckpt' if sharded else '.state'}"

    def execute_iteration(self, save_every: Optional[int] = None, t0: int = 0, return_state: bool = True):
        """One PS iteration: reweight -> train -> resample -> mutate -> commit (core.py:162-185).  Returns host copies
        of the current state like the reference unless return_state=False (the run loop does not need them: for 10^5+
        particles that copy is megabytes over PCIe per iteration)."""
        self._ensure_callbacks()
        if self.state.get_current("iter") is None:
            self._initialize_fresh()
        if save_every is not None:
            it = self.state.get_current("iter")
            if (it - t0) % int(save_every) == 0 and it != t0:
                self.save_sampler_state(self._checkpoint_path(str(it)))
        if getattr(self, "profile", False):
            return self._execute_iteration_profiled()
        weights = self.reweighter.run()
        mode_stats = self.trainer.run(weights)
        self.resampler.run(weights)
        self.mutator.run(mode_stats)
        self._update_progress_bar()
        self.state.commit_current_to_history()
        return self.state.get_current() if return_state else None

    def _execute_iteration_profiled(self):
        """Same pipeline with a device synchronisation after each phase; accumulates wall seconds in self.timing
        (diagnostics only: the synchronisations remove the overlap the normal path has)."""
        import time
        import torch
        dev = self.state.device

        def lap(name, t0):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            self.timing[name] += t1 - t0
            return t1
        t = time.perf_counter()
        weights = self.reweighter.run(); t = lap("reweight", t)
        mode_stats = self.trainer.run(weights); t = lap("train", t)
        self.resampler.run(weights); t = lap("resample", t)
        self.mutator.run(mode_stats); t = lap("mutate", t)
        self.state.commit_current_to_history(); lap("commit", t)
        return None

    def _logz_at(self, beta):
        m, s1, s2 = self.state.reweight_eval([beta])[0]
        return float(s1 * s1 / s2), float(m + np.log(s1))

    def _not_termination(self) -> bool:
        """Continue while (1 - beta >= 1e-4) or ESS(beta=1) < n_total (core.py:360-374)."""
        if self.state.get_history_length() == 0:
            return True
        beta = self.state.get_current("beta")
        if 1.0 - beta >= 1e-4:            # the first clause decides: ESS(beta = 1) -- one pass over the history -- is not needed
            return True
        ess, _ = self._logz_at(1.0)
        return ess < getattr(self, "n_total", 0)

    # ------------------------------------------------------------------------------ outputs
    def compute_posterior(self, resample=False, return_blobs=False, trim_importance_weights=True, return_logw=False,
                          ess_trim=0.99, bins_trim=1000):
        """Weighted posterior samples from the whole history (core.py:187-242)."""
        import torch
        st = self.state
        ctx = st.ctx
        ctx.use_current_stream()
        m, s1, _ = st.reweight_eval([1.0])[0]
        nh = st.n_history_global()
        w_dev = ctx.weights(1.0, m, s1)
        logw = None
        if return_logw:       # the untrimmed, normalised log-weights (core.py:233-242)
            logw = ctx.logw(1.0, nh).cpu().numpy() - (float(m + np.log(s1)) + np.log(nh))
        comm = st.comm
        blobs = st.get_history("blobs", flat=True) if (self.config.blobs_dtype is not None and st._blobs) else None
        if comm is None or not comm.active:
            # single shard: threshold, compaction, resampling and the gather into row-major (M, d) stay on the device;
            # only the M returned rows cross PCIe
            sel, m_sel, wdiv = None, ctx.size, 1.0
            if trim_importance_weights:
                thr_dev, out = ctx.trim_threshold(w_dev, ess_trim, bins_trim, sync=True)
                m_sel, wdiv = int(out[2]), float(out[1])
                sel = ctx.compact_indices(w_dev, thr_dev[0:1], m_sel)
            if resample:
                from .tools import SQRTEPS
                _, _, w_sel = ctx.posterior_rows(sel, m_sel, w=w_dev, wdiv=wdiv)
                cdf = ctx.cdf(w_sel)
                tot = float(cdf[-1].item())
                pick = ctx.resample_systematic(cdf, m_sel, np.random.random(),
                                               renorm=tot if abs(tot - 1.0) > SQRTEPS else 1.0)
                sel = pick if sel is None else ctx.index_compose(sel, pick)
                x_dev, logl_dev, _ = ctx.posterior_rows(sel, m_sel)
                weights = np.ones(m_sel) / m_sel
            else:
                x_dev, logl_dev, w_sel = ctx.posterior_rows(sel, m_sel, w=w_dev, wdiv=wdiv)
                weights = w_sel.cpu().numpy()
            x, logl = x_dev.cpu().numpy(), logl_dev.cpu().numpy()
            if blobs is not None and sel is not None:
                blobs = blobs[sel.cpu().numpy()]
        else:
            # sharded run: every rank returns the posterior over the WHOLE history (rows of all shards, rank order).
            # The trim threshold is a statistic of the global weight distribution: the library finds it without moving
            # the weights (tph_trim_threshold_global); every shard then compacts and gathers its own kept rows on the
            # device and only those rows travel.  The weights are already normalised by the global sum.
            if logw is not None:
                logw = comm.gather_rows(logw)
            sel, m_sel, wdiv = None, ctx.size, 1.0
            if trim_importance_weights:
                thr_dev, out = ctx.trim_threshold(w_dev, ess_trim, bins_trim, sync=True, global_=True)
                wdiv = float(out[1])
                m_sel = int((w_dev >= thr_dev[0]).sum().item())      # this shard's share of the kept rows
                sel = ctx.compact_indices(w_dev, thr_dev[0:1], m_sel) if m_sel else None
            if m_sel:
                x_dev, logl_dev, w_sel = ctx.posterior_rows(sel, m_sel, w=w_dev, wdiv=wdiv)
                x, logl, weights = x_dev.cpu().numpy(), logl_dev.cpu().numpy(), w_sel.cpu().numpy()
            else:
                x, logl, weights = np.empty((0, st.n_dim)), np.empty(0), np.empty(0)
            if blobs is not None and sel is not None:
                blobs = blobs[sel.cpu().numpy()]
            x, logl, weights = comm.gather_rows(x), comm.gather_rows(logl), comm.gather_rows(weights)
            if blobs is not None:
                blobs = comm.gather_rows(blobs)
            if resample:
                from .tools import SQRTEPS
                wt = torch.from_numpy(np.ascontiguousarray(weights)).to(ctx.device)
                cdf = ctx.cdf(wt)
                tot = float(weights.sum())
                idx = ctx.resample_systematic(cdf, len(weights), np.random.random(),
                                              renorm=tot if abs(tot - 1.0) > SQRTEPS else 1.0).cpu().numpy()
                x, logl = x[idx], logl[idx]
                if blobs is not None:
                    blobs = blobs[idx]
                weights = np.ones(len(idx)) / len(idx)
        out = [x, weights, logl]
        if return_blobs and blobs is not None:
            out.append(blobs)
        if return_logw:
            out.append(logw)
        return tuple(out)

    def compute_evidence(self):
        return self.state.get_current("logz"), getattr(self, "logz_err", None)

    # -------------------------------------------------------------------------- persistence
    def save_sampler_state(self, path: Union[str, Path], format: Optional[str] = None):
        """Reference layout {_current, _history, n_dim, random_state, n_total, logz_err} (core.py:249-279);
        the callbacks are not pickled (the reference's `sampler=dill.dumps(core)` entry is omitted).  `format="native"`,
        a path ending in ".ckpt", or a sharded run write the directory format of tempest_amd/checkpoint.py instead."""
        import dill
        from . import checkpoint
        path = Path(path)
        if checkpoint.wants_native(path, format, self.state.comm):
            return checkpoint.save(self, path)
        path.parent.mkdir(parents=True, exist_ok=True)
        d = self.state.to_dict()
        d["random_state"] = self.config.random_state
        d["n_total"] = getattr(self, "n_total", None)
        d["logz_err"] = getattr(self, "logz_err", None)
        d["rng"] = (self.rng.seed, self.rng.tick)
        with open(path, "wb") as f:
            dill.dump(d, f)

    def load_sampler_state(self, path: Union[str, Path]):
        """Restore current state AND history (the reference drops the history, core.py:289)."""
        import dill
        from . import checkpoint
        if checkpoint.is_native(path):
            checkpoint.load(self, path)
            defaults = {"iter": 0, "calls": 0, "beta": 0.0, "logz": 0.0, "steps": 0, "acceptance": 0.0,
                        "efficiency": 0.0, "cv": None}
            for key, val in defaults.items():
                if self.state.get_current(key) is None:
                    self.state.set_current(key, val)
            return
        with open(Path(path), "rb") as f:
            d = dill.load(f)
        self.state.update_from_dict(d)
        defaults = {"iter": 0, "calls": 0, "beta": 0.0, "logz": 0.0, "steps": 0, "acceptance": 0.0, "efficiency": 0.0,
                    "cv": None}
        for key, val in defaults.items():
            if self.state.get_current(key) is None:
                self.state.set_current(key, val)
        if "n_total" in d:
            self.n_total = d["n_total"]
        if "logz_err" in d:
            self.logz_err = d["logz_err"]
        if d.get("rng") is not None:
            self.rng.seed, self.rng.tick = d["rng"]
        elif d.get("random_state") is not None:
            np.random.seed(d["random_state"])

    # ------------------------------------------------------------------------------ helpers
    def _log_like(self, x):
        """Host-array likelihood in the reference's (logl, blobs) convention (core.py:317-358)."""
        self._ensure_callbacks()
        return self.callbacks.loglike_host(np.asarray(x))

    def _initialize_fresh(self):
        self.state.update_current({"iter": 0, "calls": 0, "beta": 0.0, "logz": 0.0})

    def _initialize_from_resume(self, path):
        self.load_sampler_state(path)
        it = self.state.get_current("iter")
        self.t0 = int(it) if it is not None else 0

    def _update_progress_bar_initial(self):
        if self.pbar is not None:
            self.pbar.update_stats(dict(beta=0.0, calls=0, ESS=int(self.config.ess_ratio * self.config.n_particles),
                                        logZ=0.0, logL=0.0, acc=0.0, steps=0, eff=0.0, K=1))

    def _update_progress_bar(self):
        if self.pbar is None or self.pbar.progress_bar.disable:
            return
        cur = self.state._current
        logl = cur["logl"]
        stats = dict(calls=cur["calls"], beta=cur["beta"], ESS=int(cur["ess"]), logZ=cur["logz"],
                     logL=float(logl.mean().item()) if logl is not None else 0.0, acc=cur["acceptance"],
                     steps=cur["steps"], eff=cur["efficiency"])
        if cur.get("cv") is not None:
            stats["CV"] = cur["cv"]
        self.pbar.update_stats(stats)

    def _get_distribute_func(self):
        pool = self.config.pool
        if pool is None:
            return map
        if isinstance(pool, int) and pool > 1:
            if getattr(self, "_pool", None) is None:       # one pool for the run, not one per call
                from multiprocess import Pool
                self._pool = Pool(pool)
            return self._pool.map
        return pool.map
