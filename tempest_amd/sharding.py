"""Particle sharding across GPUs (one process per GPU).

Each rank owns n_particles/G slots of every iteration's active set and appends them to its LOCAL history
shard; the history never moves.  The GLOBAL history order is the reference's (iteration-major, then slot), so a
sharded run is the same sampler as the one-GPU run: the same counter-based draws are mapped through the same
cumulative weights to the same history rows, and the proposal is fitted on the whole weighted history (global
trim threshold, global up-sampling, all-reduced moments and medians: csrc/modes.hip, csrc/resample.hip).
Cross-GPU traffic:
  * reweight: all-gather of the (max, s1, s2) triples per pass (merged on the device);
  * train: the small reductions of the global fit (see comm.py);
  * mutation: all-reduce of (accepted, sum alpha_c) per MCMC step;
  * resample: the selected rows travel once per iteration: written by their holders straight into the slot owners' windows
    over the peer mapping (one node, tph_resample_put_global), else by all-to-all-v (this module).
"""
import numpy as np


def resample_sharded(state, w, scheme, rng, n_local):
    """Global multinomial / systematic resampling of n_local * G slots from the sharded weighted history.
    Slot i is owned by rank i // n_local.  Every rank evaluates the position of every global draw (counter-based RNG:
    the same numbers everywhere) and looks up only those that land in its own blocks of the global cumulative weight;
    it gathers those rows and ships them to the slot owners with one all-to-all-v.  Returns this rank's (u, x, logl)
    as (d, n_local) / (n_local,) tensors."""
    import torch
    from .device import TAG_RESAMPLE, TAG_SYST
    from ._philox_host import uniform_scalar
    from .tools import SQRTEPS
    ctx, comm = state.ctx, state.comm
    G, me, d = comm.world_size, comm.rank, state.n_dim
    n_slots = n_local * G
    tick = rng.next()
    if scheme == "mult":
        cdf = ctx.cdf_global(w)
        idx = ctx.resample_select_global(cdf, n_slots, 0, rng.seed, tick, tag=TAG_RESAMPLE)
    else:
        cdf, tot = ctx.cdf_global(w, total=True)
        u0 = uniform_scalar(rng.seed, tick, TAG_SYST)
        idx = ctx.resample_select_global(cdf, n_slots, 1, rng.seed, tick, u0=u0,
                                         pscale=tot if abs(tot - 1.0) > SQRTEPS else 1.0, tag=TAG_RESAMPLE)
    if ctx.p2p_active:
        # ranks of one node: every holder writes its rows straight into the slot owner's window over the peer mapping
        # (tph_resample_put_global) -- no counts, no packing, no all-to-all, no host synchronisation
        got = ctx.resample_put_global(idx, n_local)
        if got is not None:
            return got
    slots = torch.nonzero(idx >= 0).reshape(-1)                # my outgoing slots, ascending = grouped by owner
    rows = idx[slots].contiguous()
    n_send = int(rows.numel())
    send_counts = torch.bincount(slots // n_local, minlength=G)
    # one row per claimed slot: (u, x, logl, slot inside the owner's shard) -- the slot travels with the row, so that one
    # all-to-all-v places everything (a rank receives from each sender a block in slot order, the blocks interleave)
    packed = torch.empty(2 * d + 2, max(n_send, 1), dtype=torch.float64, device=ctx.device)
    if n_send:
        ctx.gather(rows, packed[:d], packed[d:2 * d], packed[2 * d])
        packed[2 * d + 1, :n_send] = (slots % n_local).to(torch.float64)
    send = packed[:, :n_send].T.contiguous()                   # (n_send, 2d+2) rows: contiguous block per owner
    recv_counts = comm.all_to_all_counts(send_counts)
    if int(recv_counts.sum()) != n_local:
        raise RuntimeError(f"resample shuffle: rank {me} would receive {int(recv_counts.sum())} rows, expected {n_local}")
    recv = comm.all_to_all_rows(send, send_counts.tolist(), recv_counts.tolist())
    soa = torch.empty(2 * d + 1, n_local, dtype=torch.float64, device=ctx.device)
    soa[:, recv[:, 2 * d + 1].long()] = recv[:, :2 * d + 1].T
    return soa[:d], soa[d:2 * d], soa[2 * d]


def gather_rows_in_order(comm, pos, cols):
    """Rows scattered over the ranks, each with its global position `pos` (int64, a permutation of 0..M-1 over all
    ranks) -> the whole (.., M) array in position order on every rank.  cols: (k, m_local) device tensor."""
    import torch
    k = cols.shape[0]
    packed = torch.cat([pos.to(torch.float64).reshape(1, -1), cols], dim=0).T.contiguous()    # (m_local, 1 + k)
    allr, _ = comm.all_gather_v(packed)
    M = allr.shape[0]
    out = torch.empty(k, M, dtype=torch.float64, device=cols.device)
    out[:, allr[:, 0].long()] = allr[:, 1:].T
    return out
