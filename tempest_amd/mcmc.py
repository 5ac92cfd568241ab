"""MCMC mutation on the device (reference: tempest/mcmc.py).

Host control only: the per-step loop of BaseMCMCRunner.run (mcmc.py:142-208) with its proposal,
acceptance, sigma adaptation and adaptive stopping rule executed by the HIP kernels of
csrc/mutate.hip.  The two user callbacks are the only other work in a step.  Steps before the
minimum step count need no host synchronisation (the rule cannot fire earlier, mcmc.py:119-131);
afterwards one 48-byte state read per step decides whether to stop, exactly where the reference
evaluates `_check_convergence`; that read overlaps with the (speculative) proposal of the next step.
"""
from typing import Callable, Optional

import numpy as np


class PhiloxStream:
    """Host side of the counter-based RNG: a 64-bit seed and the tick handed to each RNG-consuming launch."""

    def __init__(self, seed: int):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.tick = 0

    def next(self) -> int:
        self.tick = (self.tick + 1) & 0xFFFFFFFF
        return self.tick


def _bc_flags(n_dim, periodic, reflective):
    f = np.zeros(n_dim, dtype=np.uint8)
    if periodic is not None and len(periodic):
        f[np.asarray(periodic, dtype=int)] = 1
    if reflective is not None and len(reflective):
        f[np.asarray(reflective, dtype=int)] = 2
    return f


def apply_boundary_conditions(u, periodic=None, reflective=None):
    """Wrap periodic and fold reflective coordinates into [0, 1] (mcmc.py:326-366).  Host utility with the
    same arithmetic as the device proposal kernel (`v % 1.0`; floor parity flip)."""
    out = np.array(u, dtype=np.float64, copy=True)
    flags = _bc_flags(out.shape[-1], periodic, reflective)
    for j in np.nonzero(flags == 1)[0]:
        out[..., j] = np.mod(out[..., j], 1.0)
    for j in np.nonzero(flags == 2)[0]:
        v = out[..., j]
        k = np.floor(v)
        frac = v - k
        out[..., j] = np.where(np.mod(k, 2.0) == 0.0, frac, 1.0 - frac)
    return out


def check_bounds(u, periodic=None, reflective=None):
    """True where every coordinate WITHOUT a boundary condition lies in [0, 1] (mcmc.py:369-411)."""
    u = np.asarray(u)
    strict = np.nonzero(_bc_flags(u.shape[-1], periodic, reflective) == 0)[0]
    if strict.size == 0:
        return True if u.ndim == 1 else np.ones(u.shape[0], dtype=bool)
    s = u[..., strict]
    inside = (s >= 0) & (s <= 1)
    return bool(inside.all()) if u.ndim == 1 else inside.all(axis=-1)


class DeviceMCMC:
    """One mutation run over the active set held on a GPU context."""

    def __init__(self, ctx, kernel: str, beta: float, mode_stats, log_likelihood: Callable, prior_transform: Callable,
                 n_steps: int, n_max: int, periodic=None, reflective=None, rng: Optional[PhiloxStream] = None,
                 comm=None, item0: int = 0, n_global: Optional[int] = None, progress_bar=None, verbose=True):
        import torch
        self.ctx, self.kernel, self.beta, self.modes = ctx, kernel, float(beta), mode_stats
        self.loglike, self.prior = log_likelihood, prior_transform
        self.n_steps, self.n_max = int(n_steps), int(n_max)
        self.rng = rng if rng is not None else PhiloxStream(np.random.randint(0, 2 ** 62))
        self.comm, self.item0, self.n_global = comm, int(item0), n_global
        self.pbar, self.verbose = progress_bar, verbose
        d = ctx.n_dim
        flags = _bc_flags(d, periodic, reflective)
        self.bc = torch.from_numpy(flags).to(ctx.device) if flags.any() else None
        self.sigma_0 = 2.38 / np.sqrt(d)

    def run(self, u, x, logl, assignments):
        """u, x: (d, n) device tensors (updated in place); logl (n,); assignments int32 (n,) or None.
        Returns (efficiency, acceptance, iterations, n_calls) like mcmc.py:196-208."""
        import torch
        ctx, modes, d = self.ctx, self.modes, self.ctx.n_dim
        n = u.shape[1]
        n_global = n if self.n_global is None else self.n_global
        K = modes.K
        assign = assignments if K > 1 else None
        sig0 = min(self.sigma_0, 0.99) if self.kernel == "tpcn" else self.sigma_0      # mcmc.py:222-223,298-299
        sigmas = torch.full((K,), sig0, dtype=torch.float64, device=ctx.device)
        counts = ctx.cluster_counts(assign, n, K)
        state = ctx.zeros(6)
        sums = ctx.empty(1 + K)
        up, maha_u, maha_up = ctx.empty(d, n), ctx.empty(n), ctx.empty(n)
        active = self.comm is not None and self.comm.active
        if active:
            self.comm.all_reduce_sum(counts)
        n_min = self.n_steps * d
        it, calls = 0, 0
        st = None
        state_host = torch.empty(6, dtype=torch.float64).pin_memory()
        ev = torch.cuda.Event()
        speculated = False

        def propose():
            ctx.propose(self.kernel, u, assign, modes, sigmas, self.bc, self.rng.seed, self.rng.next(), self.item0,
                        up, maha_u, maha_up)
        while True:
            it += 1
            if not speculated:
                propose()
            speculated = False
            xp = self.prior(up)                       # (d, n) SoA tensor
            lp = self.loglike(xp)                     # (n,) tensor
            calls += n_global
            ctx.accept(self.kernel, self.beta, u, x, logl, up, xp, lp, maha_u, maha_up, assign, K, modes.dof_dev,
                       self.rng.seed, self.rng.next(), self.item0, sums)
            if active:
                self.comm.all_reduce_sum(sums)
            ctx.adapt(self.kernel, sums, counts, K, n_global, self.n_steps, self.n_max, sigmas, state)
            if it >= n_min:
                # read the 48-byte step state while the NEXT step's proposal (which only needs the adapted sigma,
                # already ordered on the stream) is being generated; if the stopping rule fired it is discarded
                state_host.copy_(state, non_blocking=True)
                ev.record()
                propose()
                speculated = True
                ev.synchronize()
                st = state_host.numpy().copy()
                if self.pbar is not None and self.verbose:
                    self.pbar.update_stats({"calls": self.pbar.info.get("calls", 0) + n_global, "acc": st[3],
                                            "steps": it, "eff": st[4]})
                if st[1] != 0.0:
                    break
        return float(st[4]), float(st[3]), it, calls


def parallel_mcmc(u, x, logl, blobs, assignments, beta, mode_stats, log_likelihood, prior_transform,
                  progress_bar=None, n_steps: int = 100, n_max: int = 1000, sample: str = "tpcn", periodic=None,
                  reflective=None, verbose: bool = True):
    """Drop-in for tempest.mcmc.parallel_mcmc (mcmc.py:414-508) on host arrays: u, x (n, d), logl (n,),
    callbacks in the reference's convention (prior_transform per row, log_likelihood(x) -> (logl, blobs)).
    Returns (u, x, logl, blobs, efficiency, acceptance, iterations, n_calls)."""
    import torch
    from .tools import _ctx
    u = np.asarray(u, dtype=np.float64)
    n, d = u.shape
    ctx = _ctx(d)
    dev = ctx.device

    def prior_dev(up):
        uh = np.ascontiguousarray(up.cpu().numpy().T)
        xh = np.array([prior_transform(row) for row in uh])
        return torch.from_numpy(np.ascontiguousarray(xh.T)).to(dev)

    def like_dev(xp):
        xh = np.ascontiguousarray(xp.cpu().numpy().T)
        ll, _ = log_likelihood(xh)
        return torch.from_numpy(np.ascontiguousarray(ll, dtype=np.float64)).to(dev)

    to_soa = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(dev)  # noqa: E731
    ut, xt = to_soa(u), to_soa(x)
    lt = torch.from_numpy(np.array(logl, dtype=np.float64)).to(dev)
    at = torch.from_numpy(np.asarray(assignments, dtype=np.int32)).to(dev)
    if mode_stats.means_dev.device != dev:
        raise ValueError("mode_stats lives on another device")
    run = DeviceMCMC(ctx, "rwm" if sample == "rwm" else "tpcn", beta, mode_stats, like_dev, prior_dev, n_steps, n_max,
                     periodic, reflective, progress_bar=progress_bar, verbose=verbose)
    eff, acc, it, calls = run.run(ut, xt, lt, at)
    back = lambda t: np.ascontiguousarray(t.cpu().numpy().T)  # noqa: E731
    return back(ut), back(xt), lt.cpu().numpy(), blobs, eff, acc, it, calls
