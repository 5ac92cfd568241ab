"""Particle sharding across GPUs (one process per GPU).

Each rank owns n_particles/G slots of every iteration's active set and appends them to its LOCAL history
shard; the history never moves.  Cross-GPU traffic:
  * reweight: all-gather of the (max, s1, s2) partials (comm.merge_triples), 24 B per rank per trial beta;
  * mutation: all-reduce of (accepted, sum alpha_c) per MCMC step;
  * resample: the selected rows travel once per iteration by all-to-all-v (this module);
  * proposal fit: none -- every rank fits the proposal on its own shard.  A shard is an exchangeable 1/G
    subsample of the weighted history (slots are filled by i.i.d. / stratified draws from the global
    weights), so the per-shard median/covariance estimate the same quantities; the Metropolis correction
    uses the proposing rank's own statistics, which keeps every rank's kernel exactly invariant.
"""
import numpy as np


def resample_sharded(state, w, scheme, rng, n_local):
    """Global multinomial / systematic resampling of n_local * G slots from the sharded weighted history.
    Slot i is owned by rank i // n_local.  Every rank evaluates all global draws (counter-based RNG), keeps the
    ones that fall in its own span of the global cumulative weight, gathers those rows and ships them to the
    slot owners with one all-to-all-v.  Returns this rank's (u, x, logl) as (d, n_local) / (n_local,) tensors."""
    import torch
    from .device import TAG_RESAMPLE, TAG_SYST
    from ._philox_host import uniform_scalar
    ctx, comm = state.ctx, state.comm
    G, me, d = comm.world_size, comm.rank, state.n_dim
    n_slots = n_local * G
    cdf = ctx.cdf(w)
    tot = comm.all_gather(cdf[-1:].clone()).cpu().numpy().reshape(-1)       # per-rank weight totals, rank order
    bounds = np.concatenate([[0.0], np.cumsum(tot)])                          # identical on every rank
    tick = rng.next()
    u0 = uniform_scalar(rng.seed, tick, TAG_SYST) if scheme != "mult" else 0.0
    idx = ctx.resample_select(cdf, n_slots, 0 if scheme == "mult" else 1, rng.seed, tick, u0, bounds[me],
                              bounds[me + 1], bounds[-1], (1 if me == G - 1 else 0) | (2 if me == 0 else 0), TAG_RESAMPLE)
    slots = torch.nonzero(idx >= 0).reshape(-1)                # my outgoing slots, ascending = grouped by owner
    rows = idx[slots].contiguous()
    n_send = int(rows.numel())
    send_counts = torch.bincount(slots // n_local, minlength=G)
    packed = torch.empty(2 * d + 1, max(n_send, 1), dtype=torch.float64, device=ctx.device)
    if n_send:
        ctx.gather(rows, packed[:d], packed[d:2 * d], packed[2 * d])
    send = packed[:, :n_send].T.contiguous()                   # (n_send, 2d+1) rows: contiguous block per owner
    recv_counts = comm.all_to_all_counts(send_counts)
    if int(recv_counts.sum()) != n_local:
        raise RuntimeError(f"resample shuffle: rank {me} would receive {int(recv_counts.sum())} rows, expected {n_local}")
    recv = comm.all_to_all_rows(send, send_counts.tolist(), recv_counts.tolist())
    soa = recv.T.contiguous()                                   # (2d+1, n_local)
    return soa[:d], soa[d:2 * d], soa[2 * d]


def fit_modes_sharded(state, w, trim_ess, trim_bins, dof_fallback, rng):
    """Per-shard proposal fit (see module docstring): the single-GPU path on the local shard, with the
    local weights renormalised to sum to one and a rank-specific RNG tick."""
    import torch
    from .modes import ModeStatistics
    ctx, comm = state.ctx, state.comm
    n_h = w.numel()
    base = rng.next()
    for _ in range(comm.world_size - 1):
        rng.next()
    tick = (base + comm.rank) & 0xFFFFFFFF
    s = ctx.sum_sq_max(w)
    wl = w / float(s[0])                                         # local renormalisation (scalar scale)
    thr = ctx.trim_threshold(wl, trim_ess, trim_bins)
    cdf = ctx.cdf(wl, thr[0:1])
    counts = ctx.multinomial_counts(cdf, rng.seed, tick, kept_count=thr[2:3], factor=4, n_draw_max=4 * n_h)
    means, covs, chol, inv, winv = ctx.fit_modes(counts, None, 1, n_h)
    dof = torch.full((1,), float(dof_fallback), dtype=torch.float64, device=ctx.device)
    return ModeStatistics(None, None, None, _dev=(ctx, means, covs, chol, inv, dof, winv))
