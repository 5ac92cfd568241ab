"""Public facade: `tempest_amd.Sampler` is a drop-in for `tempest.Sampler` (tempest/sampler.py:12-406):
same constructor parameters and defaults, same methods, same read-only properties.  Everything numeric runs
on the MI355X through libtempest_hip.so; GPU-specific options are keyword-only."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

from .config import SamplerConfig
from .core import SamplerCore
from .state_manager import StateManager
from .tools import FunctionWrapper


class Sampler:
    def __init__(
        self,
        prior_transform: callable,
        log_likelihood: callable,
        n_dim: int,
        n_particles: Optional[int] = None,
        ess_ratio: float = 2.0,
        volume_variation: Optional[float] = None,
        log_likelihood_args: Optional[list] = None,
        log_likelihood_kwargs: Optional[dict] = None,
        vectorize: bool = False,
        blobs_dtype: Optional[str] = None,
        periodic: Optional[list] = None,
        reflective: Optional[list] = None,
        pool: Optional[Union[int, object]] = None,
        clustering: bool = True,
        normalize: bool = True,
        cluster_every: int = 1,
        split_threshold: float = 1.0,
        n_max_clusters: Optional[int] = None,
        sample: str = "tpcn",
        n_steps: Optional[int] = None,
        n_max_steps: Optional[int] = None,
        resample: str = "mult",
        output_dir: Optional[str] = None,
        output_label: Optional[str] = None,
        random_state: Optional[int] = None,
        *,
        device: Optional[Union[int, str]] = None,
        backend: str = "auto",
        batch_prior: Optional[bool] = None,
        graph: Optional[bool] = None,
        distributed: Optional[bool] = None,
        student_em: bool = False,
    ):
        """GPU additions (keyword-only):
        device      -- GPU index (default: current torch device; LOCAL_RANK under torchrun).
        backend     -- "torch": callbacks receive torch-ROCm FP64 tensors (x is an (n, n_dim) strided view of the
                       SoA buffer) and return tensors; "numpy": callbacks receive NumPy arrays staged through the
                       host; "auto" (default) probes.
        batch_prior -- True if prior_transform accepts an (n, n_dim) batch (probed when None).
        graph       -- True: with device callbacks the MCMC step (proposal, both callbacks, Metropolis update,
                       adaptation) is captured once as a hipGraph and replayed; the callbacks must then be pure device
                       functions of their argument (no host synchronisation, no Python side effects per call -- a
                       callback that cannot be captured falls back with a warning).  False: launch every step's
                       kernels one by one.  None (default): graph when the shard is small enough for the step to be
                       launch-bound (n_particles_per_gpu * n_dim <= 2^19).
        student_em  -- True: every proposal mode's (mu, Sigma, nu) comes from the Student-t EM of tempest/student.py:66-116 with a
                       working degrees-of-freedom update, started from the default estimator (an EXTENSION: the reference's
                       own loop returns its start values with nu = inf -> dof_fallback; see tempest_amd/student.py).  One GPU only.
        distributed -- shard the particles over the ranks of the initialised torch.distributed group
                       (default: yes if a group is initialised); n_particles is the GLOBAL count."""
        wrapped = FunctionWrapper(log_likelihood, log_likelihood_args, log_likelihood_kwargs) \
            if (log_likelihood_args or log_likelihood_kwargs) else log_likelihood
        config = SamplerConfig(
            prior_transform=prior_transform, log_likelihood=wrapped, n_dim=n_dim, n_particles=n_particles,
            ess_ratio=ess_ratio, volume_variation=volume_variation, log_likelihood_args=log_likelihood_args,
            log_likelihood_kwargs=log_likelihood_kwargs, vectorize=vectorize, blobs_dtype=blobs_dtype,
            periodic=periodic, reflective=reflective, pool=pool, clustering=clustering, normalize=normalize,
            cluster_every=cluster_every, split_threshold=split_threshold, n_max_clusters=n_max_clusters,
            sample=sample, n_steps=n_steps, n_max_steps=n_max_steps, resample=resample, output_dir=output_dir,
            output_label=output_label, random_state=random_state, device=device, backend=backend,
            batch_prior=batch_prior, graph=graph, student_em=student_em)
        comm = None
        if distributed is not False:
            from .comm import Comm
            c = Comm()
            if c.active:
                comm = c
            elif distributed:
                raise ValueError("distributed=True needs an initialised torch.distributed process group")
        if config.student_em and comm is not None:
            # rejected here, before any rank has sampled the prior: inside the first Trainer.run the other ranks would already
            # be waiting at a collective
            raise ValueError("student_em=True is not available on a sharded run (the Student-t EM is a one-GPU extension)")
        # reserve the history of a typical run (~40-60 PS iterations) up front: growing it later means hipMalloc + copy +
        # hipFree of gigabytes in the middle of the run (measured: one 230 ms stall at the 16M -> 32M row step)
        world = comm.world_size if (comm is not None and comm.active) else 1
        state = StateManager(n_dim, device=device, comm=comm, capacity_hint=64 * max(1, config.n_particles // world))
        self._core = SamplerCore(config, state)
        self.state = state
        # process start-up (first-use code-object loads, copy path, device context) in parallel, behind the construction
        from . import _warm
        self._core._warm = _warm.start(config, state)

    @property
    def startup_breakdown(self) -> dict:
        """What the parallel start-up of this Sampler's process cost (tempest_amd/_warm.py); empty for later Samplers."""
        w = getattr(self._core, "_warm", None)
        return w.breakdown() if w is not None else {}

    # ------------------------------------------------------------------------------- methods
    def run(self, n_total: int = 4096, progress: bool = True, resume_state_path: Union[str, Path, None] = None,
            save_every: Optional[int] = None):
        """Run Persistent Sampling until beta = 1 and the ESS of the whole history reaches n_total."""
        return self._core.run_sampling(n_total=n_total, progress=progress, resume_state_path=resume_state_path,
                                       save_every=save_every)

    def sample(self, save_every: Optional[int] = None, t0: int = 0, *, return_state: bool = True) -> dict:
        """One iteration (reweight, train, resample, mutate, commit); returns host copies of the current state
        (the reference's contract).  return_state=False skips that device-to-host copy."""
        return self._core.execute_iteration(save_every=save_every, t0=t0, return_state=return_state)

    def posterior(self, resample: bool = False, return_blobs: bool = False, trim_importance_weights: bool = True,
                  return_logw: bool = False, ess_trim: float = 0.99, bins_trim: int = 1000) -> tuple:
        """(x, weights, logl[, blobs][, logw]) over the whole history, NumPy, C-contiguous, copies."""
        return self._core.compute_posterior(resample=resample, return_blobs=return_blobs,
                                            trim_importance_weights=trim_importance_weights, return_logw=return_logw,
                                            ess_trim=ess_trim, bins_trim=bins_trim)

    def evidence(self) -> tuple:
        """(logZ, None): the reference never computes an error estimate (core.py:151,244-247)."""
        return self._core.compute_evidence()

    def results(self):
        return self.state.compute_results()

    def save_state(self, path: Union[str, Path], *, format: Optional[str] = None):
        """format=None: the reference's dill layout, unless `path` ends in ".ckpt" or the run is sharded over several
        GPUs -- then (and with format="native") a checkpoint directory of raw per-shard dumps (checkpoint.py)."""
        self._core.save_sampler_state(Path(path), format=format)

    def load_state(self, path: Union[str, Path]):
        self._core.load_sampler_state(Path(path))

    # ---------------------------------------------------------------------------- properties
    n_dim = property(lambda self: self._core.config.n_dim)
    n_particles = property(lambda self: self._core.config.n_particles)
    ess_ratio = property(lambda self: self._core.config.ess_ratio)
    volume_variation = property(lambda self: self._core.config.volume_variation)
    n_steps = property(lambda self: self._core.config.n_steps)
    n_max_steps = property(lambda self: self._core.config.n_max_steps)
    n_total = property(lambda self: getattr(self._core, "n_total", None))
    resample = property(lambda self: self._core.config.resample)
    clustering = property(lambda self: self._core.config.clustering)
    vectorize = property(lambda self: self._core.config.vectorize)
    output_dir = property(lambda self: self._core.config.output_dir)
    output_label = property(lambda self: self._core.config.output_label)
    random_state = property(lambda self: self._core.config.random_state)
    periodic = property(lambda self: self._core.config.periodic)
    reflective = property(lambda self: self._core.config.reflective)
    beta = property(lambda self: self.state.get_current("beta"))
    logz = property(lambda self: self.state.get_current("logz"))
    ess = property(lambda self: self.state.get_current("ess"))
    cv = property(lambda self: self.state.get_current("cv"))
