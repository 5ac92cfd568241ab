"""Mutation step (reference: tempest/steps/mutate.py:76-200): fresh prior draws while beta = 0
(with the +-inf-likelihood repair and its logZ correction), MCMC on the device otherwise."""
import numpy as np

from ..mcmc import DeviceMCMC


class Mutator:
    def __init__(self, state, prior_transform, log_likelihood, pbar=None, n_particles: int = 256, n_dim: int = 1,
                 n_steps: int = 10, n_max_steps: int = 1000, sampler: str = "tpcn", periodic=None, reflective=None,
                 have_blobs: bool = False, rng=None, device_callbacks=None, graph=None):
        """`prior_transform(u)` / `log_likelihood(x) -> (logl, blobs)` follow the reference's conventions unless
        `device_callbacks=(prior_dev, like_dev)` is given: SoA tensor -> SoA tensor / (n,) tensor (set by SamplerCore)."""
        self.state = state
        self.prior_transform = prior_transform
        self.log_likelihood = log_likelihood
        self.pbar = pbar
        self.n_particles = n_particles
        self.n_dim = n_dim
        self.n_steps = n_steps
        self.n_max_steps = n_max_steps
        self.sampler = sampler
        self.periodic = periodic
        self.reflective = reflective
        self.have_blobs = have_blobs
        self.rng = rng
        self.device_callbacks = device_callbacks
        self.graph = graph              # None: replay the MCMC step as a hipGraph when the callbacks are device functions
        self._engines = {}

    def _rng(self):
        if self.rng is None:
            from ..mcmc import PhiloxStream
            self.rng = PhiloxStream(np.random.randint(0, 2 ** 62))
        return self.rng

    def _callbacks(self):
        if self.device_callbacks is not None:
            return self.device_callbacks
        import torch
        dev = self.state.device

        def prior_dev(up):      # reference convention: one row at a time on the host (mcmc.py:157)
            uh = np.ascontiguousarray(up.cpu().numpy().T)
            xh = np.array([self.prior_transform(r) for r in uh])
            return torch.from_numpy(np.ascontiguousarray(xh.T)).to(dev)

        def like_dev(xp, return_blobs=False):
            xh = np.ascontiguousarray(xp.cpu().numpy().T)
            ll, blobs = self.log_likelihood(xh)
            ll = torch.from_numpy(np.ascontiguousarray(ll, dtype=np.float64)).to(dev)
            return (ll, blobs) if return_blobs else ll
        return prior_dev, like_dev

    def run(self, mode_stats) -> None:
        import torch
        st = self.state
        ctx = st.ctx
        ctx.use_current_stream()
        rng = self._rng()
        prior_dev, like_dev = self._callbacks()
        comm = st.comm
        active = comm is not None and comm.active
        n = self.n_particles                      # rows held by this rank
        n_global = n * (comm.world_size if active else 1)
        item0 = n * (comm.rank if active else 0)
        beta = st.get_current("beta")
        if self.have_blobs and active:
            raise NotImplementedError("likelihood blobs live on the host of the rank that computed them and are not moved "
                                      "by the sharded resampling: run blobs on one GPU")
        if beta == 0.0:
            u = ctx.empty(self.n_dim, n)
            ctx.prior_draw(u, rng.seed, rng.next(), item0)
            x = prior_dev(u)
            blobs = None
            if self.have_blobs:
                logl, blobs = like_dev(x, return_blobs=True)
                logl = logl.clone()
            else:
                logl = like_dev(x).clone()
            x = x.clone() if x.data_ptr() == u.data_ptr() else x
            calls = st.get_current("calls") + n_global
            if blobs is not None:     # the repaired rows take their donor's blob as well (mutate.py:135-136)
                stats, src = ctx.inf_repair(u, x, logl, rng.seed, rng.next(), item0, return_src=True)
                blobs = np.asarray(blobs)[src.cpu().numpy()]
            else:
                stats = ctx.inf_repair(u, x, logl, rng.seed, rng.next(), item0)    # in place; finite rows untouched
            if active:
                comm.all_reduce_sum(stats)
            n_fin, n_tot = stats.cpu().numpy()
            st.update_current({"u": u, "x": x, "logl": logl,
                               "assignments": torch.zeros(n, dtype=torch.int32, device=st.device), "calls": calls,
                               "steps": 1, "acceptance": 1.0, "efficiency": 1.0}, copy=False)
            if blobs is not None:
                st.set_current("blobs", blobs)
            if n_fin < n_tot:     # logZ correction for the prior mass without finite likelihood (mutate.py:144-148)
                with np.errstate(divide="ignore"):
                    st.set_current("logz", st.get_current("logz") + float(np.log(n_fin / n_tot)))
            return

        blobs = st.get_current("blobs") if self.have_blobs else None      # mutate.py:151-155
        u, x, logl = st.dev("u"), st.dev("x"), st.dev("logl")
        adapter = getattr(self.device_callbacks[0], "__self__", None) if self.device_callbacks is not None else None
        on_device = getattr(adapter, "backend", None) == "torch"
        plugin = getattr(adapter, "hip_plugin", None)      # both callbacks are HIP device functions: fused step
        # auto: graphs pay when a step's kernels are launch-bound (small shards); at >= ~5e5 coordinates per shard the
        # host keeps ahead of the GPU anyway and the engine's copy-in/copy-out (1-2 %) is not recovered
        want = self.graph if self.graph is not None else n * self.n_dim <= (1 << 19)
        # device callbacks: device-side step control (mcmc.StepEngine); blobs are host data that follow every accepted move,
        # so they take the step-by-step path
        engines = self._engines if on_device and blobs is None else None
        if blobs is not None:
            plugin = None
        run = DeviceMCMC(ctx, "rwm" if self.sampler == "rwm" else "tpcn", beta, mode_stats, like_dev, prior_dev,
                         self.n_steps, self.n_max_steps, self.periodic, self.reflective, rng=rng, comm=comm,
                         item0=item0, n_global=n_global, progress_bar=self.pbar, engines=engines, graph=bool(want),
                         plugin=plugin)
        efficiency, acceptance, steps, mcmc_calls = run.run(u, x, logl, st.dev("assignments"), blobs=blobs)
        st.update_current({"efficiency": efficiency, "acceptance": acceptance, "steps": steps,
                           "calls": st.get_current("calls") + mcmc_calls})
        if blobs is not None:
            st.set_current("blobs", run.blobs)
