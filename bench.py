"""Headline benchmark: particle-mutation-steps/s of the persistent-SMC loop on 10-D Rosenbrock
(BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full Persistent Sampling iteration (reweight -> train -> resample -> mutate -> commit; the beta = 0
prior-draw iterations that initialise a run come first and are neither warm-up nor timed steps) of
`tempest_amd.Sampler` on the README Rosenbrock target (coefficient 10, prior U(-10,10)^10) with BASELINE config 4's
1 048 576 particles -- at every N: the configuration fits one GPU (7 GB of history at termination), so at N=1 it is the
workload as it stands, and under torchrun the SAME 1 048 576 particles are sharded over the N ranks (131 072 per GPU at
N=8, config 4 as BASELINE.json states it): "scaling": "strong".  `value` = particle-mutation-steps (sum over the timed
iterations of MCMC steps x global particles, the reference's `calls` bookkeeping, mcmc.py:89) / wall time, user
likelihood included, inputs resident in HBM.  `--particles-per-gpu P` overrides the shard size (global = P x N).

Also in the JSON line:
  roofline      the reweight reduction kernel (north_star's named kernel) on a 1.07 GB synthetic history
                (outside the 256 MB Infinity Cache): algorithmic 16 B per historical particle /
                HIP-event average launch duration, against 8 TB/s nominal AND against the streaming-read / copy ceilings
                measured on this box in this process (`measured_peak`); the 2.6e6- and 1.05e7-row points of SURVEY 8d.
  mutation_only pms/s inside the Mutator, from per-phase timings of the untimed tail of the run.
  weak_scaling  (N > 1) the same protocol with 1 048 576 particles PER GPU.
  cpu_baseline  the NumPy oracle sampler ("port") on the host cores, bounded sample, same target.
  hip_callbacks the same run with the callbacks compiled into the Metropolis kernel (tempest_amd.HipCallbacks).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ANALYTIC_LOGZ = 5 * np.log(np.pi / np.sqrt(10.0)) - 10 * np.log(20.0)      # -29.9901
HBM_PEAK_GBS = 8000.0                                                       # MI355X_MICROARCH.md


def rosenbrock_torch(x):
    return -(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0).sum(dim=1)


def rosenbrock_numpy(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)


# The same target as HIP device functions (tempest_amd.HipCallbacks): evaluated inside the Metropolis kernel.
ROSENBROCK_HIP = """
__device__ void prior_transform(const double* u, double* x) {
#pragma unroll
  for (int j = 0; j < N_DIM; ++j) x[j] = 20.0 * u[j] - 10.0;
}
__device__ double log_likelihood(const double* x) {
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < N_DIM; j += 2) {
    double a = x[j] * x[j] - x[j + 1], b = x[j] - 1.0;
    s += 10.0 * a * a + b * b;
  }
  return -s;
}
"""


def prior20(u):
    return 20 * u - 10


def _synthetic_history(ctx, n_rows):
    """logl ~ -chi2_10 * (1 + t/T), T = 64, generated in blocks to bound host memory."""
    T = 64
    rs = np.random.RandomState(0)
    n_t = np.full(T, n_rows // T, dtype=np.int64)
    n_t[-1] += n_rows - n_t.sum()
    logl = np.empty(n_rows)
    off = 0
    for t in range(T):
        logl[off:off + n_t[t]] = -rs.chisquare(10, size=n_t[t]) * (1 + t / T)
        off += n_t[t]
    ctx.history_load(None, None, logl, np.linspace(0, 1, T) ** 2, -np.linspace(0, 30, T), n_t)


def reweight_roofline(device, n_rows, other_rows=(2_621_440, 10_485_760)):
    """HIP-event timing of the reduction kernel on synthetic histories of n_rows (16 B each), beside the box's own
    streaming ceilings measured in this process."""
    import torch
    from tempest_amd.device import HipContext
    ctx = HipContext(1, device)
    _synthetic_history(ctx, n_rows)
    ms = [ctx.reweight_time(0.37, 1, 20) for _ in range(5)]
    avg_ms = float(np.median(ms))
    m, s1, s2 = ctx.reweight_eval([0.37])[0]
    algo_bytes = 16.0 * n_rows
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
    # the box's own ceilings on the same number of bytes: a read-only stream (what the reduction is) and a copy
    read_ms = float(np.median([ctx.membw_time(0, 2 * n_rows, 20) for _ in range(3)]))
    copy_ms = float(np.median([ctx.membw_time(1, n_rows, 20) for _ in range(3)]))
    read_gbs, copy_gbs = algo_bytes / (read_ms * 1e-3) / 1e9, algo_bytes / (copy_ms * 1e-3) / 1e9
    fp64 = ctx.fp64_tflops()
    points = []
    for rows in other_rows:                      # SURVEY 8d's two smaller histories (inside / around the 256 MB Infinity Cache)
        _synthetic_history(ctx, rows)
        t = float(np.median([ctx.reweight_time(0.37, 1, 50) for _ in range(5)]))
        points.append({"history_rows": rows, "bytes": 16 * rows, "avg_launch_ms": round(t, 5),
                       "achieved": round(16.0 * rows / (t * 1e-3) / 1e9, 1), "unit": "GB/s",
                       "note": "working set inside the 256 MB Infinity Cache: not an HBM figure" if 16 * rows < 256e6 else ""})
    ctx.close()
    torch.cuda.empty_cache()
    traffic, traffic_source = None, None
    import glob
    import re
    # counter traffic of this kernel on this history: the NEWEST round's profiled run (profiles/rNN_reweight_pmc.json)
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reweight_pmc.json")),
                      key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True):
        if traffic is not None:
            break
        name = os.path.basename(pmc)
        try:
            rec = json.load(open(pmc))
            if rec.get("n_rows") == n_rows:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/" + name + ": rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE of "
                                  "this kernel on this history in that round's profiled run, NOT counters of this run")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "k_reweight_reduce<1, 8>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(avg_ms, 5), "history_rows": n_rows,
            "measured_peak": {"read_GBs": round(read_gbs, 1), "copy_GBs": round(copy_gbs, 1),
                              "what": "16-B non-temporal streaming read / copy kernels of the library (tph_bench_membw_time) over the "
                                      "same 1.07 GB in this process"},
            "frac_of_measured": round(achieved / read_gbs, 4), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
            "fp64_vector_tflops_measured": round(fp64, 2),
            "other_points": points, "check_ess": float(s1 * s1 / s2)}


def cpu_baseline(budget_s=5.0):
    """Oracle (NumPy port of the reference algorithm) on the host: 10-D Rosenbrock, N=4096, as many PS
    iterations as fit the time budget (at least the 3 warm-up + 2 annealing ones)."""
    from oracle.sampler import OracleSampler
    try:
        import threadpoolctl
        threads = max(d.get("num_threads", 1) for d in threadpoolctl.threadpool_info()) if threadpoolctl.threadpool_info() else 1
    except Exception:
        threads = 1
    n_cpu = 4096
    s = OracleSampler(prior20, rosenbrock_numpy, 10, n_cpu, seed=0)
    t0 = time.perf_counter()
    it = 0
    while (time.perf_counter() - t0 < budget_s or it < 5) and it < 60:
        s.sample()
        it += 1
    wall = time.perf_counter() - t0
    return {"value": round(s.pms / wall, 1), "unit": "particle-mutation-steps/s", "cores": int(threads), "kind": "port",
            "sample": f"oracle.sampler.OracleSampler, 10-D Rosenbrock, N={n_cpu}, first {it} PS iterations "
                      f"({s.pms} particle-mutation-steps in {wall:.1f} s; host has {os.cpu_count()} cores, "
                      f"NumPy/BLAS threads={threads})",
            "phase_seconds": {k: round(v, 2) for k, v in s.timing.items()}}


def main():
    # everything except the final JSON line goes to stderr: RCCL prints a version banner on stdout at communicator
    # creation, and the driver expects exactly one line there
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=1048576, help="GLOBAL particles: BASELINE config 4's 1 048 576")
    ap.add_argument("--particles-per-gpu", type=int, default=0,
                    help="override: this many per rank (global = P x N); default: --particles / N (strong scaling)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the extra weak-scaling run (1 048 576 per GPU)")
    ap.add_argument("--roofline-rows", type=int, default=67_108_864)      # 1.07 GB of (logl, logmix)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-second-run", action="store_true", help="skip the warm repeat of the whole run (whole_run_warm)")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="MCMC step replayed as a hipGraph (Sampler(graph=...)); auto = the library's size rule")
    ap.add_argument("--no-finish", action="store_true", help="skip running on to termination for logZ")
    ap.add_argument("--no-hip-callbacks", action="store_true",
                    help="skip the second run with the callbacks compiled into the step (tempest_amd.HipCallbacks)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--clustering", action="store_true",
                    help="run with the Sampler's default clustering=True (SURVEY 8d fixes clustering=False for config 4: the "
                         "headline line is produced without this flag; rehearsals use it to walk the clustered code path)")
    ap.add_argument("--roofline-only", action="store_true", help="only the reweight-kernel microbench (for rocprofv3 --pmc)")
    a = ap.parse_args()
    if a.roofline_only:
        emit({"roofline": reweight_roofline(int(os.environ.get("LOCAL_RANK", "0")), a.roofline_rows)})
        return

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    # TEMPEST_AMD_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- walks the N > 1 control flow (sharding, the peer-to-peer
    # layer between processes, max-over-ranks timing, the extra legs) on a one-GPU box; its numbers mean nothing
    rehearsal = os.environ.get("TEMPEST_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("TEMPEST_AMD_FORCE_COMM") == "1"
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # a rank that never arrives must end the job, not hang it: the device-side exchange gives up after 30 s (a step of
        # this workload takes milliseconds on every rank), the host's wait for a step record after 60 s, a process-group
        # collective after 120 s (RCCL's first collective builds its rings); main() turns any of them into exit code 1
        os.environ.setdefault("TEMPEST_AMD_P2P_TIMEOUT", "30")
        os.environ.setdefault("TEMPEST_AMD_STEP_TIMEOUT", "60")
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("TEMPEST_AMD_PG_TIMEOUT", "120")))
        if rehearsal:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    dev = torch.device("cuda", local_rank)

    import tempest_amd as tp
    if a.particles_per_gpu > 0:
        n_local = a.particles_per_gpu
    else:
        if a.particles % world:
            raise SystemExit(f"--particles {a.particles} is not divisible by {world} ranks")
        n_local = a.particles // world
    n_global = n_local * world
    n_total = 4 * n_global                                                  # n_total = 4 N

    def make(callbacks, n_glob):
        return tp.Sampler(callbacks[0], callbacks[1], 10, n_particles=n_glob, vectorize=True, clustering=bool(a.clustering),
                          random_state=a.seed, backend="torch", batch_prior=True, device=local_rank,
                          graph={"auto": None, "on": True, "off": False}[a.graph])
    t_make0 = time.perf_counter()
    s = make((prior20, rosenbrock_torch), n_global)

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    def timed(s):
        """W untimed + K timed PS iterations; (seconds [max over ranks], steps per timed iteration, betas)."""
        n_global = s._core.config.n_particles
        import gc
        # initialisation: the beta = 0 iterations draw the first ensembles from the prior (no MCMC steps, nothing to count);
        # they are never part of the W warm-up or K timed steps, so that any W, K >= 1 measures mutation work
        n_init = 0
        while n_init == 0 or s.state.get_current("beta") == 0.0:
            s.sample(return_state=False)
            n_init += 1
            if s.state.get_current("beta") > 0.0:
                break
        for _ in range(a.warmup):
            s.sample(return_state=False)
        sync()
        it0 = len(s.state._scalars["steps"])
        gc.collect()
        gc.disable()          # as timeit does: a generation-2 collection (tens of ms with torch loaded) is not the workload
        t0 = time.perf_counter()
        for _ in range(a.steps):
            s.sample(return_state=False)
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, np.asarray(s.state._scalars["steps"][it0:]), np.asarray(s.state._scalars["beta"][it0:])

    sync()
    t_run0 = t_make0          # `whole_run` is timed from BEFORE the Sampler's construction: its parallel start-up begins there
    dt, steps_t, beta_t = timed(s)
    startup = s.startup_breakdown
    comm_block = None
    if use_dist:
        # what the sharded run actually did between the ranks during everything up to here (initialisation, warm-up, timed steps)
        st = s.state.ctx.comm_stats()
        its = max(1, len(s.state._scalars["steps"]))
        p2p_here = 1 if s.state.ctx.p2p_active else 0
        flag = torch.tensor([p2p_here, -p2p_here], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # min over ranks of p2p and of -p2p: all on, all off, or mixed
        p2p_all, p2p_any = bool(flag[0].item() == 1), bool(flag[1].item() == -1)
        comm_block = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "p2p_active_on_every_rank": p2p_all, "p2p_active_on_some_rank": p2p_any,
                      "small_collectives_path": ("peer-to-peer exchange kernels (HIP IPC inboxes over xGMI)" if p2p_all else
                                                 "process group (the ranks agreed that the peer mapping is not usable)"),
                      "per_iteration_on_rank0": {"p2p_exchanges": round(st["p2p_exchanges"] / its, 1),
                                                 "process_group_collectives_by_the_library": round(st["callback_collectives"] / its, 1),
                                                 "process_group_bytes": round(st["callback_bytes"] / its),
                                                 "shuffle_rows": round(st["shuffle_rows"] / its), "shuffle_bytes": round(st["shuffle_bytes"] / its)},
                      "iterations_counted": its,
                      "note": "library counters (tph_comm_stats) of rank 0 over the initialisation, warm-up and timed iterations; "
                              "shuffle_rows = this rank's slots refilled per iteration, (world-1)/world of them cross xGMI on average"}
        assert p2p_all == p2p_any, "the ranks disagree about the peer-to-peer layer"      # tph_comm_p2p_attach agrees by construction
        # every rank's own account of the peer-to-peer attach (the self-test runs inside tph_comm_p2p_attach): on, or why not
        reasons = [None] * world
        dist.all_gather_object(reasons, {"rank": rank, "device": local_rank, "p2p": bool(p2p_here),
                                         "reason": getattr(s.state.comm, "p2p_reason", "?")})
        comm_block["p2p_by_rank"] = reasons
    pms = float(np.sum(steps_t[beta_t > 0])) * n_global
    value = pms / dt

    n_before = len(s.state._scalars["steps"]) - len(steps_t)
    extra = {"timed_mcmc_steps": int(np.sum(steps_t[beta_t > 0])), "timed_iterations": int(a.steps),
             "timed_iteration_range_of_run": [n_before + 1, n_before + len(steps_t)],
             "beta_range": [float(beta_t.min()), float(beta_t.max())],
             "reweight_evals_per_iteration": round(s._core.reweighter.n_evals / max(1, len(s.state._scalars["beta"])), 1)}
    if not a.no_finish:
        # run on to the reference's stopping rule for the evidence (untimed)
        core = s._core
        core.n_total = n_total
        guard = 0
        # the tail is untimed by `value`: it is run with a device synchronisation after every phase, which gives the
        # per-phase split (mutation-only throughput) at the price of a few tens of microseconds per iteration
        core.profile = True
        tail0 = len(s.state._scalars["steps"])
        for k in core.timing:
            core.timing[k] = 0.0
        while core._not_termination() and guard < 400:
            s.sample(return_state=False)
            guard += 1
        core.profile = False
        _, logz = core._logz_at(1.0)
        sync()
        t_run = time.perf_counter() - t_run0
        # A digest of what the run arrived at: the evidence and the final ensemble (u and logl of ALL ranks in slot order), bit for
        # bit.  With the canonical partition (csrc/common.h: tph_part) a run on N GPUs is the same floating-point computation as
        # the run on one, so the SCALE lines (N = 2, 4, 8) and the BENCH line (N = 1) of one seed must carry the SAME digest.
        import hashlib
        u_fin, l_fin = s.state.get_current("u"), s.state.get_current("logl")
        if use_dist:
            u_fin, l_fin = s.state.comm.gather_rows(u_fin), s.state.comm.gather_rows(l_fin)
        hsh = hashlib.sha256(np.ascontiguousarray(u_fin).tobytes())
        hsh.update(np.ascontiguousarray(l_fin).tobytes())
        sched = hashlib.sha256(np.asarray(s.state._scalars["logz"], dtype=np.float64).tobytes())
        sched.update(np.asarray(s.state._scalars["beta"], dtype=np.float64).tobytes())
        from tempest_amd.device import vshards_for
        extra["digest"] = {"logz_hex": float(logz).hex(), "ensemble_sha256": hsh.hexdigest(), "schedule_sha256": sched.hexdigest(),
                           "virtual_shards": vshards_for(n_global) if n_global % 256 == 0 else 1,
                           "note": "equal across --gpus N for every N that divides virtual_shards (same seed, same particle count): "
                                   "the sharded run is bitwise the one-GPU run (tests/test_distributed.py::test_world_size_invariance_is_bitwise)"}
        del u_fin, l_fin
        tail_steps = np.asarray(s.state._scalars["steps"][tail0:]); tail_beta = np.asarray(s.state._scalars["beta"][tail0:])
        if len(tail_steps) and core.timing["mutate"] > 0:
            tail_pms = float(np.sum(tail_steps[tail_beta > 0])) * n_global
            extra["mutation_only"] = {"value": tail_pms / core.timing["mutate"], "unit": "particle-mutation-steps/s",
                                      "iterations": [tail0 + 1, tail0 + len(tail_steps)],
                                      "phase_seconds": {k: round(v, 4) for k, v in core.timing.items()},
                                      "whole_iteration_value": tail_pms / sum(core.timing.values()),
                                      "note": "untimed tail of the run (after the K timed iterations), one device "
                                              "synchronisation after every phase; mutate = time inside Mutator.run incl. the "
                                              "user's callbacks"}
        if use_dist and comm_block is not None and len(tail_steps):
            # where an iteration's time goes on EVERY rank (the tail's per-phase split, ms per iteration): a phase whose time does
            # not shrink with the shard -- replicated work, a slow exchange -- shows here, rank by rank
            keys = list(core.timing)
            mine = torch.tensor([core.timing[k] * 1e3 / len(tail_steps) for k in keys], dtype=torch.float64,
                                device="cpu" if rehearsal else dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            comm_block["phase_ms_per_iteration_by_rank"] = {k: [round(float(t[i].item()), 3) for t in allr] for i, k in enumerate(keys)}
        all_steps = np.asarray(s.state._scalars["steps"]); all_beta = np.asarray(s.state._scalars["beta"])
        extra["whole_run"] = {"value": float(np.sum(all_steps[all_beta > 0])) * n_global / t_run, "unit": "particle-mutation-steps/s",
                              "seconds": t_run, "iterations": int(len(all_beta)),
                              "note": "Sampler construction + first sample() to the reference's stopping rule on this rank, "
                                      "one-time costs (history allocation, code-object loads, callback probing) included; the "
                                      "tail of this run carries one device synchronisation per phase (mutation_only)",
                              "startup_breakdown": startup}
        if not a.no_second_run:
            # the same run once more in this process (new Sampler, same seed): what Sampler.run() costs once the process has
            # loaded its kernels -- the first iteration of the first run spends ~0.5 s in torch's lazy loading of the code
            # objects behind the user's eager callbacks (measured: 0.27 s for five elementwise kernels) and the first rocPRIM sort
            s_w = make((prior20, rosenbrock_torch), n_global)
            sync()
            t_w0 = time.perf_counter()
            s_w.run(n_total=n_total, progress=False)              # the public entry point, to the reference's stopping rule
            logz_w = s_w.evidence()[0]
            sync()
            t_w = time.perf_counter() - t_w0
            if use_dist:
                tt = torch.tensor([t_w], dtype=torch.float64, device="cpu" if rehearsal else dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_w = float(tt.item())
            w_steps = np.asarray(s_w.state._scalars["steps"]); w_beta = np.asarray(s_w.state._scalars["beta"])
            extra["whole_run_warm"] = {"value": float(np.sum(w_steps[w_beta > 0])) * n_global / t_w,
                                       "unit": "particle-mutation-steps/s", "seconds": t_w, "iterations": int(len(w_beta)),
                                       "same_logz_as_first_run": bool(logz_w == logz),
                                       "note": "a second complete run in the same process: Sampler(...).run(n_total) + evidence() on a "
                                               "fresh Sampler with the same seed, no per-phase synchronisation -- the whole "
                                               "run without the process's one-time costs"}
            del s_w
            torch.cuda.empty_cache()
        extra.update({"logz": logz, "logz_abs_err_vs_analytic": abs(logz - ANALYTIC_LOGZ),
                      "analytic_logz": float(ANALYTIC_LOGZ), "iterations_total": len(s.state._scalars["beta"]),
                      "reference_logz_ensemble_N1000": {"mean": -29.804, "std": 0.115, "seeds": 16,
                                                        "source": "tests/golden/ref_ensembles.json"}})
    out = {"metric": "particle-mutation-steps/s (whole job), 10-D Rosenbrock", "value": value,
           "unit": "particle-mutation-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
           "scaling": "strong" if a.particles_per_gpu <= 0 else "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo, not a measurement)",
           "config": {"workload": f"rosenbrock10d_n{n_global} (BASELINE config 4: 10-D Rosenbrock, {n_global} particles "
                                  f"sharded over {world} GPU(s), {n_local} per GPU)",
                      "n_dim": 10, "particles_per_gpu": n_local, "particles_global": n_global, "sample": "tpcn",
                      "resample": "mult", "clustering": bool(a.clustering), "n_total": n_total, "step": "one PS iteration"}}
    out.update(extra)
    if comm_block is not None:
        out["comm"] = comm_block
    out["config"]["callbacks"] = "torch-ROCm tensor callbacks (the drop-in contract: prior20, rosenbrock_torch)"
    if not a.no_hip_callbacks:
        # same workload, same seed, same protocol, with the two callbacks written as HIP device functions and compiled
        # into the Metropolis kernel: a step is 4 launches instead of ~16.  Reported beside `value`, never as `value`.
        try:
            cb = tp.HipCallbacks(ROSENBROCK_HIP, 10)
            del s
            torch.cuda.empty_cache()
            s2 = make((cb.prior_transform, cb.log_likelihood), n_global)
            dt2, st2, bt2 = timed(s2)
            pms2 = float(np.sum(st2[bt2 > 0])) * n_global
            out["hip_callbacks"] = {"value": pms2 / dt2, "unit": "particle-mutation-steps/s", "ms_per_step": 1e3 * dt2 / a.steps,
                                    "timed_mcmc_steps": int(np.sum(st2[bt2 > 0])),
                                    "same_schedule_as_value_run": bool(np.array_equal(st2, steps_t)),
                                    "note": "tempest_amd.HipCallbacks: prior/likelihood as HIP device functions fused "
                                            "into the Metropolis kernel (optional extension; not the headline)"}
            del s2
            torch.cuda.empty_cache()
        except Exception as e:       # no hipcc on the box, ...
            out["hip_callbacks"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    if world > 1 and a.particles_per_gpu <= 0 and not a.no_weak:
        # the weak-scaling figure beside the headline: 1 048 576 particles PER GPU, same protocol
        try:
            s = None
            torch.cuda.empty_cache()
            s3 = make((prior20, rosenbrock_torch), a.particles * world)
            dt3, st3, bt3 = timed(s3)
            out["weak_scaling"] = {"value": float(np.sum(st3[bt3 > 0])) * a.particles * world / dt3,
                                   "unit": "particle-mutation-steps/s", "particles_per_gpu": a.particles,
                                   "particles_global": a.particles * world, "ms_per_step": 1e3 * dt3 / a.steps,
                                   "timed_mcmc_steps": int(np.sum(st3[bt3 > 0]))}
            del s3
            torch.cuda.empty_cache()
        except Exception as e:
            out["weak_scaling"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        if not a.no_roofline:
            out["roofline"] = reweight_roofline(local_rank, a.roofline_rows)
        if not a.no_cpu_baseline and world == 1:       # the CPU baseline belongs to the N = 1 line only
            out["cpu_baseline"] = cpu_baseline()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


def _main_or_die():
    """Any failure -- a peer that never arrived, a collective that timed out, a device error -- ends THIS rank with exit code 1
    at once (os._exit: no interpreter shutdown that could wait on a dead communicator, and never a re-exec)."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


if __name__ == "__main__":
    _main_or_die()
