"""Resampling step: draw the next active set from the weighted history
(reference: tempest/steps/resample.py:52-99; tempest/tools.py:178-228)."""
import numpy as np


def _as_device_weights(weights, ctx):
    import torch
    dev = getattr(weights, "dev", None)
    if dev is not None:
        return dev
    return torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float64)).to(ctx.device)


class Resampler:
    """Multinomial ("mult", the reference's default) or systematic ("syst") selection of n_particles
    history rows, then a gather of u, x, logl into the current state."""

    def __init__(self, state, n_particles: int, resample: str = "syst", clusterer=None, clustering: bool = True,
                 have_blobs: bool = False, rng=None):
        self.state = state
        self.n_particles = n_particles
        self.resample = resample
        self.clusterer = clusterer
        self.clustering = clustering
        self.have_blobs = have_blobs
        self.rng = rng

    def _rng(self):
        if self.rng is None:
            from ..mcmc import PhiloxStream
            self.rng = PhiloxStream(np.random.randint(0, 2 ** 62))
        return self.rng

    def run(self, weights) -> None:
        import torch
        st = self.state
        n = self.n_particles
        beta = st.get_current("beta")
        if beta == 0.0:       # warm-up: the mutator draws fresh prior samples (resample.py:69-72)
            st.set_current("assignments", torch.zeros(n, dtype=torch.int32, device=st.device), copy=False)
            return
        ctx = st.ctx
        ctx.use_current_stream()
        rng = self._rng()
        w = _as_device_weights(weights, ctx)
        comm = st.comm
        if comm is not None and comm.active:
            if self.have_blobs:
                raise NotImplementedError("likelihood blobs live on the host of the rank that computed them and are not "
                                          "moved by the sharded resampling: run blobs on one GPU")
            from ..sharding import resample_sharded
            u, x, logl = resample_sharded(st, w, self.resample, rng, n)
        else:
            cdf = ctx.cdf(w)
            if self.resample == "mult":
                idx = ctx.resample_multinomial(cdf, n, rng.seed, rng.next())
            else:
                # tools.py:214-217: renormalise when |sum w - 1| > sqrt(eps); one uniform for the whole comb
                from ..tools import SQRTEPS
                from ..device import TAG_SYST
                tot = float(cdf[-1].item())
                renorm = tot if abs(tot - 1.0) > SQRTEPS else 1.0
                u0 = _host_uniform(rng, TAG_SYST)
                idx = ctx.resample_systematic(cdf, n, u0, renorm=renorm)
            d = st.n_dim
            u, x, logl = ctx.empty(d, n), ctx.empty(d, n), ctx.empty(n)
            ctx.gather(idx, u, x, logl)
            if self.have_blobs:      # host data: the same history rows (resample.py:77,98-99)
                st.set_current("blobs", st.get_history("blobs", flat=True)[idx.cpu().numpy()])
        if self.clustering and self.clusterer is not None:
            assign = self.clusterer.predict_device(u.contiguous(), st.ctx)
        else:
            assign = torch.zeros(logl.shape[0], dtype=torch.int32, device=st.device)
        st.update_current({"u": u, "x": x, "logl": logl, "assignments": assign}, copy=False)


def _host_uniform(rng, tag):
    """One U[0,1) from the Philox stream, evaluated on the host (same bits as the device generator)."""
    from .._philox_host import uniform_scalar
    return uniform_scalar(rng.seed, rng.next(), tag)
