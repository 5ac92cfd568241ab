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


class RegimeOptions:
    """Debugging switches of the d > 16 proposal regime (TEMPEST_AMD_STAGED / _SCREEN / _BLK_MFMA / _BLK_FAN / _SM_LANES): read from
    the environment ONCE, when an engine is built -- never in the step path."""
    __slots__ = ("walker_ok", "screen", "blk_mfma", "fan", "sm_lanes", "pin")

    def __init__(self, walker_ok=True, screen=True, blk_mfma=True, fan=1, sm_lanes=0, pin=""):
        self.walker_ok, self.screen, self.blk_mfma, self.fan, self.sm_lanes = walker_ok, screen, blk_mfma, int(fan), int(sm_lanes)
        # TEMPEST_AMD_REGIME=screened: every d > 16 step of a one-mode run through the screened batches, whatever the probe says.
        # Their proposals are the FP64 row walker's bit for bit and a particle's arithmetic never depends on its neighbours, so a
        # run pinned there is bitwise the same on any number of GPUs (the adaptive rule sends a step to the matrix-core rounds by
        # the RANK's own list lengths, and those kernels agree with the batches to rounding only)
        self.pin = pin

    @classmethod
    def from_env(cls, env=None):
        import os
        env = os.environ if env is None else env
        return cls(walker_ok=env.get("TEMPEST_AMD_STAGED", "1") != "0", screen=env.get("TEMPEST_AMD_SCREEN", "1") != "0",
                   blk_mfma=env.get("TEMPEST_AMD_BLK_MFMA", "1") != "0", fan=int(env.get("TEMPEST_AMD_BLK_FAN", "1")),
                   sm_lanes=int(env.get("TEMPEST_AMD_SM_LANES", "0")), pin=env.get("TEMPEST_AMD_REGIME", ""))


# Crossovers of the redraw probe (mean attempts per particle of a step) between the d > 16 proposal kernels.  One row per
# dimension band (the first whose n_dim_min <= n_dim):  (n_dim_min, up, down, cap, floor) --
#   up:    the blocked rounds are left when THEIR probe, the geometric estimate n / (n - first-attempt failures), reaches it;
#   down:  they are taken up again when the TRUE mean reported by the screened batches / the row walker falls below it
#          (the hard particles near a wall pull the true mean above the estimate: hence two numbers);
#   cap, floor: at most `cap` rounds, and only while the expected failure list still holds `floor` particles.
# All fitted on one MI355X with tools/regime_sweep.py; `source` names the sweep a row came from.
REGIME_THRESHOLDS = {
    "screened": {"source": "profiles/r04_regime_sweep.jsonl, profiles/r04_regime_sweep_fan.jsonl (fanned-out list rounds); the crossover "
                           "re-measured with the LDS-staged rounds of round 5: profiles/r05_regime_sweep.jsonl (unchanged)",
                 "bands": ((64, 3.0, 5.0, 6, 64.0), (33, 3.4, 5.5, 8, 64.0), (17, 8.0, 13.0, 12, 64.0))},
    "walker": {"source": "profiles/r03_propose_d50_d100.json (screen off: FP64 row walker, multi-lane straggler pass)",
               "bands": ((64, 4.5, 8.0, 24, 24576.0), (17, 3.5, 5.0, 24, 24576.0))},
    # several modes (matrix-core rounds over mode-pure tiles <-> per-mode screened batches / multi-lane kernel): leave at, return below
    "several_modes": {"source": "profiles/r04_regime_sweep_fan.jsonl", "up": 13.0, "down": 8.0},
    # multi-lane kernel: matrices from global memory (a quarter of the LDS) above, LDS-staged below
    "unstaged": {"source": "profiles/r02_propose_d50_d100.json", "up": 8.0, "down": 4.0},
    # a kernel switch retires the captured graph of the step: from the third switch in a row that comes within `recent` readings of
    # the one before, the next has to wait (4, 8, ... 64 probe readings) -- a probe that sits on a threshold and flips every step
    # costs a bounded number of captures (test_regime_rule_does_not_oscillate)
    "dwell": {"first": 4, "max": 64, "recent": 8},
}


def regime_band(table, n_dim):
    for row in REGIME_THRESHOLDS[table]["bands"]:
        if n_dim >= row[0]:
            return row[1:]
    return REGIME_THRESHOLDS[table]["bands"][-1][1:]


class StepEngine:
    """One MCMC step -- proposal, the two user callbacks, Metropolis update, sigma adaptation -- as a replayable
    hipGraph over persistent device buffers.

    A step at BASELINE config 4's shard size (131 072 particles, d = 10) is ~15 launches of 5-50 us kernels: issued one
    by one from Python the host cannot keep the GPU busy.  Everything that changes from step to step (RNG tick, stop
    flag) and from one PS iteration to the next (beta, tick base, proposal modes, the active set itself) lives in device
    memory -- the step-control block of tempest_hip.h and the buffers below -- so the graph is captured once per
    (n, K) and replayed for every later step of the run.  Steps launched past the stopping rule are no-ops on the
    device, so the host may run one step ahead of the 64-byte state read that tells it when to stop.

    With a communicator the per-rank sums are all-reduced between the Metropolis kernel and tph_adapt: by the library's
    peer-to-peer exchange kernel when that is attached (tph_comm_p2p_*; part of the step and of its graph), else by the
    host through the process group between the replay and an eagerly launched tph_adapt.  The user callbacks must be pure device functions of their argument
    (they are traced once); `graph=False` keeps the step-by-step launch path."""

    SLOTS = 64          # mailbox ring: the host never runs more than a few steps ahead of the record it waits for
    _epoch = 0          # version counter of the mode statistics handed to the library (TPH_OPT_MODES_EPOCH)

    def __init__(self, ctx, kernel, n, K, has_assign, bc, log_likelihood, prior_transform, seed, item0, n_global,
                 n_steps, n_max, comm_active, use_graph=True, plugin=None):
        import torch
        from types import SimpleNamespace
        self.plugin = plugin
        d = ctx.n_dim
        self.ctx, self.kernel, self.n, self.K, self.bc = ctx, kernel, n, K, bc
        self.loglike, self.prior = log_likelihood, prior_transform
        self.seed, self.item0, self.n_global = seed, item0, n_global
        self.n_steps, self.n_max, self.comm_active = n_steps, n_max, comm_active
        # with the library's peer-to-peer exchange attached, the all-reduce of the step's sums is a kernel on the ctx stream:
        # it belongs to the step (and to its graph) like every other launch; otherwise the host calls the process group
        # (this rank's shard sums -- vl (1 + K) doubles, vl = its virtual shards of the canonical partition -- must fit one slot)
        from .device import vshards_for
        world = max(1, int(round(n_global / n))) if n else 1
        vl = vshards_for(n * world) // world if (n * world) % 256 == 0 and vshards_for(n * world) % world == 0 else 1
        self.inline_reduce = bool(comm_active and ctx.p2p_active and 8 * vl * (1 + K) <= 32768)   # one exchange slot (p2p.h)
        self.up, self.maha_u, self.maha_up = ctx.empty(d, n), ctx.empty(n), ctx.empty(n)
        self.u = self.logl = self.assign = self.modes = None
        if use_graph:       # fixed-address copies of everything a captured step reads or writes
            self.u, self.logl = ctx.empty(d, n), ctx.empty(n)
            self.assign = torch.empty(n, dtype=torch.int32, device=ctx.device) if has_assign else None
            self.modes = SimpleNamespace(K=K, means_dev=ctx.empty(K, d), chol_dev=ctx.empty(K, d, d),
                                         winv_dev=ctx.empty(K, d, d), dof_dev=ctx.empty(K))
        self.has_assign = has_assign
        self.sigmas, self.counts, self.sums = ctx.empty(K), ctx.empty(K), ctx.zeros(1 + K)
        self.partials = ctx.empty(((n + 255) // 256) * (1 + K))     # fixed address: the library's scratch may move
        # deferred Metropolis update: tph_accept records the decisions here, the NEXT tph_propose moves the accepted
        # proposals into place (its FP64 work hides the copy; in place it was a bandwidth-bound kernel of its own)
        self.pending = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
        from .device import STEP_STATE_LEN
        self.ctl = ctx.zeros(STEP_STATE_LEN)
        self.ctl_host = torch.zeros(STEP_STATE_LEN, dtype=torch.float64).pin_memory()
        self.unstaged = False          # d > 16 proposal kernel without LDS-staged matrices (redraw-dominated steps)
        self.blocked = 0               # d > 16: rounds of the blocked kernel (attempts in lockstep) before the straggler pass; 0 = off
        self.staged, self.sm_lanes = False, 0      # d > 16: row-walker kernel for redraw-dominated steps, its lanes per particle (log2)
        self._opts = RegimeOptions.from_env()      # the regime's debugging switches, read here and nowhere in the step path
        import os
        env = os.environ.get("TEMPEST_AMD_STEP_TIMEOUT")
        self._step_timeout = float(env) if env else None      # seconds a step's record may take (None: until the stream goes idle)
        self._since, self._dwell, self._quick = 1 << 30, 0, 0      # readings since the last kernel switch; readings the next must wait; quick switches in a row
        self._screened = self._opts.screen and d <= 112
        if (d > 16 and self._screened and self._opts.walker_ok and ((K == 1 and not has_assign) or (1 < K <= 64 and has_assign))):
            # Nothing is known about the redraw rate before an engine's first steps: they run as screened batches, whose time is
            # flat in it (0.4-3 ms), instead of through the multi-lane kernel, whose time is not (30 ms per launch from the prior
            # at 131 072 x 100-D: three such launches were 2 % of the config-5 shard's run); the probe of step 2 picks the regime.
            self.staged = True
        # written by tph_adapt, polled here: the ring of step records and, behind it, the last record of a run made in ONE launch
        self._mail_all = torch.zeros(self.SLOTS + 1, 8, dtype=torch.float64).pin_memory()
        self.mailbox = self._mail_all[:self.SLOTS]
        self.mailbox_np = self.mailbox.numpy()
        self._run_bufs = None
        self.use_graph, self.graph, self.graph_error = bool(use_graph), None, None
        self._keep, self.runs, self._own = None, 0, None
        self._retired_graphs = []

    def load(self, u, x, logl, assign, modes, beta, tick_base, sigma_init, counts):
        """Start a run: active set, proposal modes and the step-control block.  A graph needs fixed addresses, so
        the data is copied into the engine's persistent buffers; without one the caller's tensors are used as they are."""
        self.runs += 1
        if self.use_graph:
            if self._own is None:
                self._own = (self.u, self.logl, self.assign, self.modes)
            self.u, self.logl, self.assign, self.modes = self._own
            self.u.copy_(u); self.logl.copy_(logl)
            if self.assign is not None:
                self.assign.copy_(assign)
            m = self.modes
            m.means_dev.copy_(modes.means_dev.reshape(m.means_dev.shape)); m.chol_dev.copy_(modes.chol_dev.reshape(m.chol_dev.shape))
            winv = getattr(modes, "winv_dev", None)
            if winv is None:            # mode statistics built outside ModeStatistics: L^-1 from the factors
                winv = torch.linalg.inv(modes.chol_dev.reshape(m.chol_dev.shape))
            m.winv_dev.copy_(winv.reshape(m.winv_dev.shape)); m.dof_dev.copy_(modes.dof_dev.reshape(m.dof_dev.shape))
        else:
            self.u, self.logl, self.assign, self.modes = u, logl, assign, modes
        if self.ctx.n_dim > 16:        # new mode statistics: the library rebuilds its blocked copies of L and L^-1 once
            from .device import OPT_BLOCKED, OPT_ML_UNSTAGED, OPT_MODES_EPOCH, OPT_SM_LANES, OPT_STAGED_REDRAW
            StepEngine._epoch += 1
            self.ctx.set_option(OPT_MODES_EPOCH, StepEngine._epoch)
            # the proposal regime is an option of the ctx, which engines of other shapes share: re-assert this engine's
            self.ctx.set_option(OPT_BLOCKED, int(self.blocked))
            self.ctx.set_option(OPT_ML_UNSTAGED, 1 if self.unstaged else 0)
            self.ctx.set_option(OPT_STAGED_REDRAW, 1 if self.staged else 0)
            self.ctx.set_option(OPT_SM_LANES, self.sm_lanes)
            from .device import OPT_SCREEN
            self.ctx.set_option(OPT_SCREEN, 1 if self._screened else 0)
        self.sigmas.fill_(sigma_init)
        self.pending.zero_()
        self.counts.copy_(counts)
        self.mailbox_np[:, 7] = -1.0          # no record yet (the device is idle or running no-op steps: see step())
        h = self.ctl_host
        h.zero_()
        h[6], h[7] = float(beta), float(tick_base)
        self.ctl.copy_(h, non_blocking=True)

    def _enqueue(self):
        """The launches of one step on the current stream (ticks are offsets: the device adds base + 2 * steps done)."""
        ctx = self.ctx
        if (self.plugin is not None and (not self.comm_active or self.inline_reduce)
                and self.plugin.can_fuse_step(self.K, self.assign is not None, self.n)):
            # proposal, callbacks and Metropolis update in ONE kernel of the user's plugin (hipcallbacks.py); with ranks only
            # when tph_adapt exchanges the sums itself (otherwise the host needs them between two launches)
            from .device import KERNEL_ID
            self.plugin.step(KERNEL_ID[self.kernel], self.u, self.logl, self.maha_u, self.modes, self.sigmas, self.bc,
                             self.seed, 1, 2, self.item0, self.ctl, self.partials)
            self._adapt(fold=True)
            return None, None
        ctx.propose(self.kernel, self.u, self.assign, self.modes, self.sigmas, self.bc, self.seed, 1, self.item0,
                    self.up, self.maha_u, self.maha_up, ctl=self.ctl, pending=self.pending)
        # the block partials of the Metropolis kernel are summed inside tph_adapt (one launch less per step), which on a
        # sharded run also exchanges the shard sums with the peers (tph_comm_p2p_*); without that exchange the ranks are combined
        # between the two launches (step(): tph_accept_sums_global)
        sums = None
        if self.plugin is not None:       # callbacks compiled into the Metropolis kernel (hipcallbacks.py)
            from .device import KERNEL_ID
            xp = lp = None
            self.plugin.accept(KERNEL_ID[self.kernel], 0.0, self.u, None, self.logl, self.up, self.maha_u, self.maha_up,
                               self.assign, self.K, self.modes.dof_dev, self.seed, 2, self.item0, sums,
                               ctl=self.ctl, partials=self.partials, pending=self.pending)
        else:
            xp = self.prior(self.up)
            lp = self.loglike(xp)
            ctx.accept(self.kernel, 0.0, self.u, None, self.logl, self.up, xp, lp, self.maha_u, self.maha_up,
                       self.assign, self.K, self.modes.dof_dev, self.seed, 2, self.item0, sums, ctl=self.ctl,
                       partials=self.partials, pending=self.pending)
        if not self.comm_active or self.inline_reduce:
            self._adapt(fold=True)
        return xp, lp

    def _adapt(self, fold=False):
        self.ctx.adapt(self.kernel, self.sums, self.counts, self.K, self.n_global, self.n_steps, self.n_max,
                       self.sigmas, self.ctl, mailbox=self.mailbox, partials=self.partials if fold else None, n=self.n)

    def can_run_all(self):
        """The whole run in one launch (HipCallbacks.run: every step, adaptation and stopping rule inside one cooperative kernel)?"""
        return (self.plugin is not None and not self.comm_active
                and self.plugin.can_run(self.K, self.assign is not None, self.n))

    def run_all(self):
        """Launch the run loaded by load() as ONE kernel and wait for its last step's record; None if the device refused the
        launch (nothing ran: step as usual)."""
        import torch
        from .device import KERNEL_ID
        if self._run_bufs is None:
            tiles = (self.n + 255) // 256
            self._run_bufs = (self.ctx.empty(4 * tiles), torch.zeros(4, dtype=torch.int32, device=self.ctx.device))
        partials2, barrier = self._run_bufs
        last = self._mail_all.numpy()[self.SLOTS]
        last[7] = -1.0
        d = self.ctx.n_dim
        if not self.plugin.run(KERNEL_ID[self.kernel], self.u, self.logl, self.maha_u, self.modes, self.sigmas, self.bc, self.seed,
                               1, 2, self.item0, self.ctl, partials2, barrier, self.counts, self.n_global, self.n_steps,
                               self.n_max, self._mail_all, self.SLOTS, self.n_max * d + 1):
            return None
        self._poll(last, None, "the run")
        if last[1] < 0.0:
            from ._lib import TempestHipError
            raise TempestHipError("MCMC run in one launch was abandoned on the device (a workgroup timed out at the grid barrier)")
        if last[1] == 0.0:
            from ._lib import TempestHipError
            raise TempestHipError("MCMC run in one launch ended without its stopping rule having fired")
        return last[:6].copy()

    def _poll(self, rec, step, what, timeout=None):
        """Spin on a pinned record until its sequence field shows `step` (None: any step); errors as in wait_record."""
        import time
        import torch
        if timeout is None:
            timeout = self._step_timeout
        arrived = (lambda: rec[7] >= 0.0) if step is None else (lambda: rec[7] == step)
        spins, t0 = 0, None
        while not arrived():
            spins += 1
            if spins & 0x3FFF == 0:
                now = time.monotonic()
                t0 = t0 or now
                if now - t0 > 0.05:
                    time.sleep(0)
                if now - t0 > 2.0 and torch.cuda.current_stream(self.ctx.device).query() and not arrived():
                    from ._lib import TempestHipError
                    raise TempestHipError(f"{what}: the stream is idle and the device never delivered the record")
                if timeout is not None and now - t0 > timeout:
                    from ._lib import TempestHipError
                    raise TempestHipError(f"{what}: no record from the device after {timeout} s")

    def wait_record(self, step, timeout=None):
        """State record (tph_adapt's state[0..5]) of step `step`, polled from the pinned mailbox.  The wait ends with an
        error only when the stream has gone IDLE without delivering the record (like tph_reweight_eval's poll) or, if a
        `timeout` in seconds is given (TEMPEST_AMD_STEP_TIMEOUT), after that long: a step may legitimately take minutes
        (an expensive likelihood, a callback that compiles on first use, a shared GPU)."""
        import time
        import torch
        if timeout is None:
            timeout = self._step_timeout
        rec = self.mailbox_np[step % self.SLOTS]
        spins, t0 = 0, None
        while rec[7] != step:
            spins += 1
            if spins & 0x3FFF == 0:
                now = time.monotonic()
                t0 = t0 or now
                if now - t0 > 0.05:
                    time.sleep(0)                 # long wait: let other Python threads run between polls
                if now - t0 > 2.0:
                    self.ctx.p2p_status()         # raises if a peer never arrived at an exchange
                if now - t0 > 2.0 and torch.cuda.current_stream(self.ctx.device).query() and rec[7] != step:
                    from ._lib import TempestHipError
                    raise TempestHipError(f"MCMC step {step}: the stream is idle and the device never delivered the step's record")
                if timeout is not None and now - t0 > timeout:
                    from ._lib import TempestHipError
                    raise TempestHipError(f"MCMC step {step}: no record from the device after {timeout} s")
        self._regime(rec[6])
        return rec[:6].copy()

    def _regime(self, mean_attempts):
        """The d > 16 proposal kernels report the mean number of attempts per particle of the step, and the host picks the
        kernel for the next steps from it (thresholds: REGIME_THRESHOLDS above).  A captured graph has its kernels baked in: when
        the rule asks for a different KERNEL than the graph holds (blocked <-> screened batches / walker <-> multi-lane; not for a
        different number of rounds), the graph is retired and the step is captured again at its next launch -- an engine captured in
        a run's redraw-dominated first iterations would otherwise walk rows for the rest of the run.  One mode:
          * a step is a few attempts per particle: the blocked rounds (propose_blkm.hip: attempts in lockstep, 16 particles per
            wave, both triangular products on the FP64 matrix cores) -- attempt 0 of everybody, further rounds over the particles
            still out of bounds (fanned out), the rest finished by a screened launch over the list.  Its probe is the geometric
            estimate n / (n - first-attempt failures);
          * most attempts are redraws: the screened batches (propose_mf.hip; screen off: the FP64 row walker, propose_sm.hip).
            Their probe is the true mean, which the hard particles near a wall pull above the geometric estimate (131 072 x
            100-D: estimate 2.8 / true 4.7; 65 536 x 50-D: 2.1 / 2.7, 4.2 / 7.2) -- hence the two thresholds per band.
        Several modes (K > 1, up to 64): the matrix-core rounds over mode-pure tiles while a step is a few attempts per particle,
        their stragglers and the redraw-dominated steps through the per-mode screened batches (screen off: the multi-lane kernel,
        un-staged while redraws dominate, LDS-staged once a step is about one attempt).
        A switch back within a few readings of the last one makes the NEXT switch wait (REGIME_THRESHOLDS["dwell"]): a probe that
        sits on a threshold and flips every step costs a bounded number of captures, not one per step."""
        if self.ctx.n_dim <= 16 or not mean_attempts > 0.0:
            return
        before = (self.blocked > 0, bool(self.staged))
        from .device import OPT_BLOCKED, OPT_ML_UNSTAGED, OPT_SCREEN, OPT_SM_LANES, OPT_STAGED_REDRAW
        opts = getattr(self, "_opts", None) or RegimeOptions.from_env()
        if opts.pin == "screened" and self.staged:
            return                     # pinned to the screened batches (RegimeOptions.pin)
        walker_ok = opts.walker_ok
        nd = self.ctx.n_dim
        screened = opts.screen and nd <= 112
        up_est, down_true, cap, floor = regime_band("screened" if screened else "walker", nd)
        multi_ok = screened and self.K <= 64 and opts.blk_mfma
        if self.K != 1:
            sm_ = REGIME_THRESHOLDS["several_modes"]
            want_blk = multi_ok and mean_attempts < (sm_["down"] if self.blocked else sm_["up"])
        elif self.blocked:             # geometric estimate
            want_blk = mean_attempts < up_est or not walker_ok
        else:                          # true mean (screened batches / row walker, or the multi-lane kernel of a run's first steps)
            want_blk = mean_attempts < down_true and (walker_ok or mean_attempts < 2.0)
        # redraw-dominated steps: the screened batches -- with several modes one mode after the other over the particles of each
        # (propose_mf.hip: tph_propose_mf_modes); the multi-lane kernel only where the screen is off or K > 64
        want_sm = (self.K == 1 or multi_ok) and not want_blk and walker_ok
        self._since = getattr(self, "_since", 1 << 30) + 1
        self._dwell = getattr(self, "_dwell", 0)
        if (want_blk, want_sm) == before:
            if self.graph is not None:
                return                 # same kernel: the graph stays (its rounds and lane groups too)
        else:
            dw = REGIME_THRESHOLDS["dwell"]
            if self._since <= self._dwell:
                return                 # a switch so soon after the last one: wait (the current kernel is correct, only not the fastest)
            # consecutive switches each within a few readings of the one before: the second is still free (a run that crosses a
            # band once each way), from the third on the wait doubles
            self._quick = getattr(self, "_quick", 0) + 1 if self._since < max(dw["recent"], 2 * self._dwell) else 0
            self._dwell = 0 if self._quick < 2 else min(dw["max"], dw["first"] << (self._quick - 2))
            self._since = 0
        rounds = 0
        if want_blk:
            # A round that still has work costs at least one tile's latency (30-45 us at 100-D) however short its list: rounds
            # pay while the list fills the chip.  Expected list after k rounds: n (1 - 1/m)^k.  Measured against one round +
            # stragglers: 262 144 x 32-D -20 ... -26 %, 131 072 x 100-D -12 ... -17 %, 65 536 x 50-D +-0 (lists too short).
            # (screened: the straggler pass is a screened launch over the list: it settles a short list in one launch's latency, a
            # long one at ~8 us per straggler and wave -- rounds pay while the expected list is more than a few hundred particles)
            f, left = max(0.0, 1.0 - 1.0 / mean_attempts), float(self.n)
            rounds = 1
            # (a matrix-core round gives its failing columns `tries` attempts in place: TPH_OPT_BLK_TRIES, 2 up to n_dim 32, 1 above)
            f_round = f ** (1 if (nd > 32 or not screened) else 2)
            fan, fan_div = screened and opts.fan != 0, {2: 1, 3: 4}.get(opts.fan, 2)
            left *= f_round                         # after round 0
            while rounds < cap and left >= floor:
                # a list round gives every listed particle G attempts side by side (TPH_OPT_BLK_FAN, propose_blkm.hip: the largest
                # power of two <= 16 with G x list <= n / 2), so the list shrinks by f_round ** G
                G = 1
                while fan and G < 16 and 2 * fan_div * G * left <= self.n:
                    G *= 2
                left *= f_round ** G
                rounds += 1
        if rounds != self.blocked:
            self.blocked = rounds
            self.ctx.set_option(OPT_BLOCKED, rounds)
            from .device import OPT_BLK_FAN
            self.ctx.set_option(OPT_BLK_FAN, opts.fan)
        lanes = opts.sm_lanes          # 0: the library sizes the lane groups (8 or 16 lanes per particle)
        if want_sm != self.staged or lanes != self.sm_lanes:
            self.staged, self.sm_lanes = want_sm, lanes
            self.ctx.set_option(OPT_STAGED_REDRAW, 1 if want_sm else 0)
            self.ctx.set_option(OPT_SM_LANES, lanes)
            self.ctx.set_option(OPT_SCREEN, 1 if screened else 0)
        un = REGIME_THRESHOLDS["unstaged"]
        want = mean_attempts > (un["down"] if self.unstaged else un["up"])      # hysteresis
        if want != self.unstaged and self.graph is None:
            self.unstaged = want
            self.ctx.set_option(OPT_ML_UNSTAGED, 1 if want else 0)
        if self.graph is not None and (self.blocked > 0, bool(self.staged)) != before:
            # retired, not destroyed: its last replay may still be in flight (and its pool holds the callbacks' outputs)
            self._retired_graphs.append((self.graph, self._keep))
            self.graph, self._keep = None, None

    def _capture(self):
        """Stream capture of one step (torch.cuda.CUDAGraph without torch.cuda.graph's empty_cache(), which would hand
        the allocator's cached blocks back to the driver in the middle of a run)."""
        import gc
        import torch
        ctx = self.ctx
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=ctx.device)
        side.wait_stream(torch.cuda.current_stream(ctx.device))
        began = False
        # No finaliser may run while the stream is capturing: one that frees device memory or synchronises (a tensor, a
        # HipContext or a CUDAGraph of an earlier Sampler that the cyclic collector happens to reach now) is an illegal call
        # during capture and aborts the process.  The collector is kept off for the duration of the capture (with it off no
        # cyclic garbage is ever finalised here; a full collection in front of every capture, as torch.cuda.graph() makes one,
        # cost 60 ms each -- 0.3 s of a 1 000-particle run's 2 s -- for nothing the switch does not already guarantee).
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(side):
                ctx.use_current_stream()          # the capturing stream
                g.capture_begin()
                began = True
                self._keep = self._enqueue()      # callback outputs live in the graph's private pool
                g.capture_end()
            self.graph = g
        except Exception as e:                    # callbacks that synchronise or branch on data cannot be captured
            if began:
                try:
                    with torch.cuda.stream(side):
                        g.capture_end()
                except Exception:
                    pass
            self.graph, self.graph_error, self.use_graph, self._keep = None, e, False, None
            import warnings
            warnings.warn(f"MCMC step could not be captured as a graph ({type(e).__name__}: {e}); "
                          "launching step by step", stacklevel=2)
        finally:
            if gc_was_on:
                gc.enable()
            ctx.use_current_stream()
            torch.cuda.current_stream(ctx.device).wait_stream(side)

    def step(self, comm=None):
        """Enqueue one full step.  An engine's first run is launched step by step (it sizes the library's scratch, warms
        the callbacks, and a shape that never repeats -- a cluster count that changes every iteration -- never pays for a
        capture); the graph is captured after the first step of its second run and replayed from then on."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
            if self.use_graph and self.runs >= 2:
                self._capture()
        if self.comm_active and not self.inline_reduce:
            self.ctx.accept_sums_global(self.partials, self.n, self.K, self.sums)
            self._adapt()


class DeviceMCMC:
    """One mutation run over the active set held on a GPU context."""

    def __init__(self, ctx, kernel: str, beta: float, mode_stats, log_likelihood: Callable, prior_transform: Callable,
                 n_steps: int, n_max: int, periodic=None, reflective=None, rng: Optional[PhiloxStream] = None,
                 comm=None, item0: int = 0, n_global: Optional[int] = None, progress_bar=None, verbose=True,
                 engines: Optional[dict] = None, graph: Optional[bool] = None, plugin=None):
        """`engines`: a dict kept by the caller across runs; given one, the steps go through a StepEngine (persistent
        buffers + device-side step control), replayed as a hipGraph unless `graph` is False."""
        import torch
        self.engines, self.graph, self.plugin = engines, graph, plugin
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

    def run(self, u, x, logl, assignments, blobs=None):
        """u, x: (d, n) device tensors (updated in place); logl (n,); assignments int32 (n,) or None.
        `blobs` (host array, one entry per particle): the likelihood's auxiliary outputs; those of accepted moves replace
        the current ones (mcmc.py:176-177) -- the log-likelihood callback is then called with return_blobs=True and must
        return (tensor, blobs); the evolved array is left in `self.blobs`.
        Returns (efficiency, acceptance, iterations, n_calls) like mcmc.py:196-208."""
        import torch
        self.blobs = None if blobs is None else np.array(blobs, copy=True)
        if blobs is not None and (self.engines is not None or self.plugin is not None):
            raise ValueError("blobs follow the step-by-step path (host callbacks): engines and plugin must be None")
        ctx, modes, d = self.ctx, self.modes, self.ctx.n_dim
        n = u.shape[1]
        n_global = n if self.n_global is None else self.n_global
        K = modes.K
        assign = assignments if K > 1 else None
        sig0 = min(self.sigma_0, 0.99) if self.kernel == "tpcn" else self.sigma_0      # mcmc.py:222-223,298-299
        counts = ctx.cluster_counts(assign, n, K)
        if self.engines is not None:
            return self._run_engine(u, x, logl, assign, K, n, n_global, sig0, counts)
        sigmas = torch.full((K,), sig0, dtype=torch.float64, device=ctx.device)
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
        partials = ctx.empty(((n + 255) // 256) * (1 + K))

        def propose():
            ctx.propose(self.kernel, u, assign, modes, sigmas, self.bc, self.rng.seed, self.rng.next(), self.item0,
                        up, maha_u, maha_up)
        while True:
            it += 1
            if not speculated:
                propose()
            speculated = False
            calls += n_global
            if self.plugin is not None:               # callbacks compiled into the Metropolis kernel (hipcallbacks.py)
                from .device import KERNEL_ID
                self.plugin.accept(KERNEL_ID[self.kernel], self.beta, u, x, logl, up, maha_u, maha_up, assign, K,
                                   modes.dof_dev, self.rng.seed, self.rng.next(), self.item0, None, partials=partials)
            else:
                xp = self.prior(up)                   # (d, n) SoA tensor
                if self.blobs is None:
                    lp = self.loglike(xp)             # (n,) tensor
                else:
                    lp, bp = self.loglike(xp, return_blobs=True)
                ctx.accept(self.kernel, self.beta, u, x, logl, up, xp, lp, maha_u, maha_up, assign, K, modes.dof_dev,
                           self.rng.seed, self.rng.next(), self.item0, None, partials=partials)
                if self.blobs is not None:
                    # an accepted row now holds its proposal, bit for bit (a proposal equal to the current point -- the
                    # redraw cap -- has the current point's blob anyway)
                    moved = (u == up).all(dim=0).cpu().numpy()
                    self.blobs[moved] = np.asarray(bp)[moved]
            if active:          # this path runs host callbacks (or blobs): the ranks are paced by them, not by the device
                ctx.accept_sums_global(partials, n, K, sums, host_paced=True)
            # one GPU: tph_adapt sums the Metropolis kernel's block partials itself (one launch less per step)
            ctx.adapt(self.kernel, sums, counts, K, n_global, self.n_steps, self.n_max, sigmas, state,
                      partials=None if active else partials, n=n)
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


    def _run_engine(self, u, x, logl, assign, K, n, n_global, sig0, counts):
        import torch
        ctx, d = self.ctx, self.ctx.n_dim
        active = self.comm is not None and self.comm.active
        if active:
            self.comm.all_reduce_sum(counts)
        key = (self.kernel, n, K, assign is not None, active, id(self.plugin))
        eng = self.engines.get(key)
        if eng is None:
            if len(self.engines) >= 2:        # at most two engines (and graph memory pools) alive at a time
                self.engines.pop(next(iter(self.engines)))
            eng = StepEngine(ctx, self.kernel, n, K, assign is not None, self.bc, self.loglike, self.prior, self.rng.seed,
                             self.item0, n_global, self.n_steps, self.n_max, active, use_graph=self.graph is not False,
                             plugin=self.plugin)
            self.engines[key] = eng
        eng.loglike, eng.prior, eng.plugin = self.loglike, self.prior, self.plugin
        tick_base = self.rng.tick
        eng.load(u, x, logl, assign, self.modes, self.beta, tick_base, sig0, counts)
        n_min = self.n_steps * d
        it, st = 0, None
        if eng.can_run_all():
            st = eng.run_all()                # every step of the run in one launch (None: refused, nothing ran)
            if st is not None:
                it = int(st[0])
                if self.pbar is not None and self.verbose:
                    self.pbar.update_stats({"calls": self.pbar.info.get("calls", 0) + it * n_global, "acc": st[3],
                                            "steps": it, "eff": st[4]})
        if st is None:
            eng.step(self.comm)
        while st is None or st[1] == 0.0:
            it += 1                           # step `it` is enqueued
            eng.step(self.comm)               # one step ahead of the read; a no-op on the device if the rule has fired
            if it < n_min:
                if it == 2 and d > 16 and eng.runs == 1 and eng.graph is None:
                    eng.wait_record(2)        # first run of this engine: learn the proposal kernel's regime after two steps
            else:                             # the rule cannot fire earlier (mcmc.py:119-131)
                st = eng.wait_record(it)
                if self.pbar is not None and self.verbose:
                    self.pbar.update_stats({"calls": self.pbar.info.get("calls", 0) + n_global, "acc": st[3],
                                            "steps": it, "eff": st[4]})
                if st[1] != 0.0:
                    break
        if eng.u is not u:                    # graph mode: the step worked on the engine's fixed-address buffers
            u.copy_(eng.u); logl.copy_(eng.logl)
        # x = prior_transform(u) is not maintained step by step (tph_accept with x = NULL writes half the bytes): one
        # evaluation for the final positions -- the same elementwise function of the same u, hence the same values
        x.copy_(self.prior(u))
        # two ticks per step + the proposal of the step that was launched ahead (same count as the step-by-step path)
        self.rng.tick = (tick_base + 2 * it + 1) & 0xFFFFFFFF
        return float(st[4]), float(st[3]), it, it * n_global


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

    def like_dev(xp, return_blobs=False):
        xh = np.ascontiguousarray(xp.cpu().numpy().T)
        ll, bl = log_likelihood(xh)
        ll = torch.from_numpy(np.ascontiguousarray(ll, dtype=np.float64)).to(dev)
        return (ll, bl) if return_blobs else ll

    to_soa = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(dev)  # noqa: E731
    ut, xt = to_soa(u), to_soa(x)
    lt = torch.from_numpy(np.array(logl, dtype=np.float64)).to(dev)
    at = torch.from_numpy(np.asarray(assignments, dtype=np.int32)).to(dev)
    if mode_stats.means_dev.device != dev:
        raise ValueError("mode_stats lives on another device")
    run = DeviceMCMC(ctx, "rwm" if sample == "rwm" else "tpcn", beta, mode_stats, like_dev, prior_dev, n_steps, n_max,
                     periodic, reflective, progress_bar=progress_bar, verbose=verbose)
    eff, acc, it, calls = run.run(ut, xt, lt, at, blobs=blobs)
    back = lambda t: np.ascontiguousarray(t.cpu().numpy().T)  # noqa: E731
    return back(ut), back(xt), lt.cpu().numpy(), run.blobs, eff, acc, it, calls


def parallel_t_preconditioned_crank_nicolson(u, x, logl, blobs, assignments, beta, mode_stats, log_likelihood, prior_transform,
                                             progress_bar=None, n_steps: int = 100, n_max: int = 1000, periodic=None,
                                             reflective=None, verbose: bool = True):
    """tempest.mcmc.parallel_t_preconditioned_crank_nicolson (mcmc.py:511-589): parallel_mcmc with the tpCN kernel."""
    return parallel_mcmc(u, x, logl, blobs, assignments, beta, mode_stats, log_likelihood, prior_transform,
                         progress_bar=progress_bar, n_steps=n_steps, n_max=n_max, sample="tpcn", periodic=periodic,
                         reflective=reflective, verbose=verbose)


def parallel_random_walk_metropolis(u, x, logl, blobs, assignments, beta, mode_stats, log_likelihood, prior_transform,
                                    progress_bar=None, n_steps: int = 1000, n_max: int = 10000, periodic=None,
                                    reflective=None, verbose: bool = True):
    """tempest.mcmc.parallel_random_walk_metropolis (mcmc.py:592-676): parallel_mcmc with the RWM kernel (note the
    reference's larger defaults for n_steps / n_max)."""
    return parallel_mcmc(u, x, logl, blobs, assignments, beta, mode_stats, log_likelihood, prior_transform,
                         progress_bar=progress_bar, n_steps=n_steps, n_max=n_max, sample="rwm", periodic=periodic,
                         reflective=reflective, verbose=verbose)
