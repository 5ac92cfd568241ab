"""Cross-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  Particles are sharded; the history never moves.  Traffic:
  * small reductions issued by libtempest_hip itself through the two collectives attached with `Comm.attach`
    (tph_comm_attach): the (max, s1, s2) reweight triples (all-gather, merged on the device), the counters of the
    global percentile select and the moment / histogram sums of the global proposal fit (all-reduce), the
    per-iteration block totals of the global cumulative weight (all-gather);
  * per MCMC step: all-reduce SUM of (accepted, sum alpha_c);
  * once per iteration: the resample shuffle (all-to-all-v of the selected rows), and with clustering the gathered
    working set of the mixture fit (all-gather-v).
With world_size == 1 and no forced communicator every method is a no-op / identity.
"""
import os

import numpy as np


def _P2P_DTYPES():
    import torch
    return (torch.float64, torch.int64, torch.int32)


def merge_triples_host(parts):
    """Merge per-rank reweight partials [(m, s1, s2)] -> global, per trial beta.
    parts: array (G, nb, 3).  logsumexp-style: M = max m_g ; s1 = sum s1_g e^{m_g-M} ; s2 = sum s2_g e^{2(m_g-M)}."""
    parts = np.asarray(parts, dtype=np.float64)
    m = parts[:, :, 0]
    M = np.max(m, axis=0)
    with np.errstate(invalid="ignore", over="ignore"):
        f = np.exp(m - M[None, :])
        f = np.where(np.isneginf(m) & np.isneginf(M)[None, :], 0.0, f)
        s1 = np.sum(parts[:, :, 1] * f, axis=0)
        s2 = np.sum(parts[:, :, 2] * f * f, axis=0)
    return np.stack([M, s1, s2], axis=1)


def attach_loopback(ctx, nbytes: int = 64 << 20, p2p: bool = False):
    """A one-rank communicator without any process group: all-reduce = identity, all-gather = copy.  Routes a single
    GPU through every sharded code path of the library (tests; the global entry points must then reproduce the plain
    ones up to summation order).  `p2p`: also attach the peer-to-peer small-message collectives (world 1: the exchange
    kernel runs against this rank's own inbox)."""
    import ctypes as C
    import torch
    from . import _lib
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=ctx.device)
    sizes = {0: 8, 1: 8, 2: 4}

    def allreduce(_user, off, count, dtype, op):
        return 0

    def allgather(_user, soff, roff, count, dtype):
        nb = count * sizes[dtype]
        buf[roff:roff + nb].copy_(buf[soff:soff + nb])
        return 0

    ar, ag = _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather)
    _lib.check(ctx.lib.tph_comm_attach(ctx._ctx, 0, 1, C.c_void_p(buf.data_ptr()), buf.numel(), ar, ag, None),
               "tph_comm_attach")
    ctx._comm_keep = (buf, ar, ag)
    if p2p and not ctx.p2p_attach([ctx.p2p_export()]):
        raise RuntimeError("peer-to-peer collectives failed their self-test")
    return buf


class Comm:
    """Thin wrapper over an initialised torch.distributed process group (or a single process)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.world_size = self._dist.get_world_size(group) if self._dist else 1
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self._p2p_ctx = None          # weakref to the DeviceContext whose library carries the small collectives itself
        self.p2p_reason = "not attached"      # why the peer-to-peer layer is on or off on this rank (bench.py prints it per rank)

    @property
    def active(self):
        # TEMPEST_AMD_FORCE_COMM=1 routes a single rank through the sharded code path and its collectives too
        # (used to exercise the RCCL calls on a one-GPU box)
        return self.world_size > 1 or (self._dist is not None and os.environ.get("TEMPEST_AMD_FORCE_COMM") == "1")

    @property
    def _stage_on_host(self):
        """gloo (the CPU-test backend) cannot move device tensors for every collective: stage through the host."""
        return self._dist is not None and self._dist.get_backend(self.group) == "gloo"

    # ------------------------------------------------------------------ collectives for libtempest_hip
    def attach(self, ctx, nbytes: int = 64 << 20):
        """Hand the library its two collectives (tph_comm_attach): they act in place on a device staging block owned
        here, addressed by byte offset, and are ordered with the current stream (RCCL work is enqueued behind it and the
        stream waits for it; gloo stages through the host).  Returns the staging tensor."""
        import ctypes as C
        import torch
        from . import _lib
        if not self.active:
            return None
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=ctx.device)
        dts = {0: (torch.float64, 8), 1: (torch.int64, 8), 2: (torch.int32, 4)}
        ops = {0: self._dist.ReduceOp.SUM, 1: self._dist.ReduceOp.MAX, 2: self._dist.ReduceOp.MIN}
        stage = self._stage_on_host
        dist, group, world = self._dist, self.group, self.world_size

        def view(off, count, dtype):
            dt, sz = dts[dtype]
            return buf[off:off + count * sz].view(dt)

        def allreduce(_user, off, count, dtype, op):
            try:
                t = view(off, count, dtype)
                if stage:
                    h = t.cpu()
                    dist.all_reduce(h, op=ops[op], group=group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=ops[op], group=group)
                return 0
            except Exception:            # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 1

        def allgather(_user, soff, roff, count, dtype):
            try:
                send, recv = view(soff, count, dtype), view(roff, count * world, dtype)
                if stage:
                    h = send.cpu()
                    parts = [torch.empty_like(h) for _ in range(world)]
                    dist.all_gather(parts, h, group=group)
                    recv.copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(recv, send, group=group)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        ar, ag = _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather)
        _lib.check(ctx.lib.tph_comm_attach(ctx._ctx, self.rank, world, C.c_void_p(buf.data_ptr()), buf.numel(), ar, ag,
                                           None), "tph_comm_attach")
        ctx._comm_keep = (buf, ar, ag)         # the library holds raw pointers to all three
        self._p2p_ctx = None
        if os.environ.get("TEMPEST_AMD_P2P", "1") != "0":
            self._attach_p2p(ctx)
        else:
            self.p2p_reason = "switched off (TEMPEST_AMD_P2P=0)"
        return buf

    def _attach_p2p(self, ctx):
        """Ranks of one node: let the library exchange its small messages (reweight triples, per-step sums, block totals,
        moments) through peer-mapped device memory -- one kernel on the ctx stream per collective, no framework call
        (tph_comm_p2p_*).  Every rank reaches the same verdict: the handles travel through the process group, and the
        library agrees on the self-test through the all-reduce attached above.  TEMPEST_AMD_P2P=0 turns it off."""
        import socket
        import weakref
        export_error = None
        try:
            handle = ctx.p2p_export()
        except Exception as e:
            handle, export_error = None, f"{type(e).__name__}: {e}"[:200]
        mine = (socket.gethostname(), ctx.device.index, handle)
        everyone = [None] * self.world_size
        if self.world_size > 1:
            self._dist.all_gather_object(everyone, mine, group=self.group)
        else:
            everyone = [mine]
        same_node = len({h for h, _, _ in everyone}) == 1
        if not same_node:
            self.p2p_reason = "off: the ranks are on different hosts (" + ", ".join(sorted({h for h, _, _ in everyone})) + ")"
            return False
        if any(h is None for _, _, h in everyone):
            missing = [r for r, (_, _, h) in enumerate(everyone) if h is None]
            self.p2p_reason = (f"off: rank(s) {missing} could not export an inbox"
                               + (f" (here: {export_error})" if export_error else ""))
            return False
        if self.world_size > 16:
            self.p2p_reason = "off: more than 16 ranks"
            return False
        if ctx.p2p_attach([h for _, _, h in everyone]):
            self._p2p_ctx = weakref.ref(ctx)
            self.p2p_reason = "on: inboxes mapped, 48-exchange self-test passed on every rank"
            return True
        self.p2p_reason = "off: the attach self-test failed on some rank (mapping or exchange pattern): " + str(ctx.last_error())[:160]
        return False

    def all_gather_v(self, t):
        """Concatenate device tensors that differ in their FIRST dimension over the ranks (rank order); every rank gets
        the whole.  Returns (tensor, counts)."""
        import torch
        if not self.active:
            return t, [int(t.shape[0])]
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        counts = [int(c) for c in self.all_gather(n).reshape(-1).cpu().tolist()]
        m = max(counts)
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        allp = self.all_gather(pad)
        return torch.cat([allp[r, : counts[r]] for r in range(self.world_size)], dim=0), counts

    def all_reduce_sum(self, t, host_paced=False):
        """In-place SUM all-reduce of a tensor (device tensor -> RCCL; CPU tensor -> gloo); small FP64 / integer device
        tensors go through the library's peer-to-peer exchange on the ctx stream when that is attached -- unless the caller
        says the ranks are `host_paced` (user callbacks running on the host between two collectives: the ranks may then be
        minutes apart, which the device-side spin of the exchange kernel must not be asked to sit out)."""
        if self.active:
            ctx = self._p2p_ctx() if (self._p2p_ctx is not None and not host_paced) else None
            if (ctx is not None and t.is_cuda and t.is_contiguous() and t.numel() * t.element_size() <= 32768
                    and t.dtype in _P2P_DTYPES() and t.device == ctx.device and ctx.p2p_active):
                ctx.allreduce_dev(t, 0)
                return t
            if t.is_cuda and self._stage_on_host:
                h = t.cpu()
                self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def sum_int(self, n):
        if not self.active:
            return int(n)
        import torch
        t = torch.tensor([int(n)], dtype=torch.int64)
        if self._dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self._dist.all_reduce(t, group=self.group)
        return int(t.item())

    def all_gather(self, t):
        """Stack equal-shape tensors of all ranks along a new leading axis."""
        import torch
        if not self.active:
            return t.unsqueeze(0)
        if self._stage_on_host:
            h = t.detach().cpu().contiguous()
            parts = [torch.empty_like(h) for _ in range(self.world_size)]
            self._dist.all_gather(parts, h, group=self.group)
            return torch.stack(parts).to(t.device)
        out = torch.empty((self.world_size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def all_to_all_counts(self, send_counts):
        """send_counts[r] rows go to rank r -> recv_counts[r] rows come from rank r (int64 tensor on the host)."""
        import torch
        sc = send_counts.detach().to(torch.int64)
        if not self.active:
            return sc.cpu()
        allc = self.all_gather(sc)                  # (G, G): row = sender
        return allc[:, self.rank].cpu()

    def merge_triples(self, part):
        """part: (nb, 3) tensor of this rank's (max, s1, s2) -> global (nb, 3) host array."""
        allp = self.all_gather(part)
        return merge_triples_host(allp.cpu().numpy())

    def all_to_all_rows(self, send, send_counts, recv_counts):
        """all-to-all-v of row blocks: `send` is (n_send, width) ordered by destination rank."""
        import torch
        recv = torch.empty((int(sum(recv_counts)), send.shape[1]), dtype=send.dtype, device=send.device)
        if not self.active:
            recv.copy_(send)
            return recv
        if self._stage_on_host:                    # gloo: exchange as a list of per-peer CPU blocks
            sh = send.detach().cpu()
            so = np.concatenate([[0], np.cumsum(send_counts)]).astype(int)
            outs = [torch.empty((int(c), send.shape[1]), dtype=send.dtype) for c in recv_counts]
            ins = [sh[so[r]:so[r + 1]].contiguous() for r in range(self.world_size)]
            self._dist.all_to_all(outs, ins, group=self.group) if self._has_gloo_all_to_all() else self._p2p_all_to_all(outs, ins)
            return torch.cat(outs).to(send.device)
        self._dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[int(c) for c in recv_counts],
                                     input_split_sizes=[int(c) for c in send_counts], group=self.group)
        return recv

    @staticmethod
    def _has_gloo_all_to_all():
        return False          # ProcessGroupGloo lacks list all_to_all: use point-to-point

    def _p2p_all_to_all(self, outs, ins):
        reqs = []
        for r in range(self.world_size):
            if r == self.rank:
                outs[r].copy_(ins[r])
                continue
            reqs.append(self._dist.isend(ins[r], r, group=self.group))
            reqs.append(self._dist.irecv(outs[r], r, group=self.group))
        for q in reqs:
            q.wait()

    def gather_rows(self, a):
        """Concatenate host arrays of all ranks along axis 0 (equal or different lengths), result on every rank."""
        import torch
        a = np.ascontiguousarray(a)
        if not self.active:
            return a
        dev = "cuda" if self._dist.get_backend(self.group) == "nccl" else "cpu"
        n = torch.tensor([a.shape[0]], dtype=torch.int64, device=dev)
        counts = self.all_gather(n).reshape(-1).cpu().numpy()
        m = int(counts.max())
        pad = np.zeros((m,) + a.shape[1:], dtype=a.dtype)
        pad[: a.shape[0]] = a
        allp = self.all_gather(torch.from_numpy(pad).to(dev)).cpu().numpy()
        return np.concatenate([allp[r, : counts[r]] for r in range(self.world_size)], axis=0)

    def barrier(self):
        if self.active:
            self._dist.barrier(group=self.group)
