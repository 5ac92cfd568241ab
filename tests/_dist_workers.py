"""Worker bodies for the multi-process tests (spawned by torch.multiprocessing)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def comm_cpu_worker(rank, world, port, out_dir):
    """gloo / CPU tensors: the collectives of tempest_amd.comm against a single-process oracle."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import ps
    from tempest_amd.comm import Comm
    _init(rank, world, port)
    comm = Comm()
    assert comm.active and comm.world_size == world and comm.rank == rank
    rs = np.random.RandomState(0)               # same data on every rank; each takes its shard
    n = 4000
    logl = -rs.chisquare(5, size=n) * 3
    bt, zt, nt = np.array([0.0, 0.2, 0.6]), np.array([0.0, -2.0, -5.0]), np.array([n // 2, n // 4, n - n // 2 - n // 4])
    cm = ps.log_mixture(logl, bt, zt, nt)
    betas = [0.0, 0.3, 1.0]
    sl = slice(rank * n // world, (rank + 1) * n // world)
    part = torch.tensor([ps.reweight_triple(logl[sl], cm[sl], b) for b in betas], dtype=torch.float64)
    merged = comm.merge_triples(part)
    for b, row in zip(betas, merged):
        m, s1, s2 = ps.reweight_triple(logl, cm, b)
        np.testing.assert_allclose(row[0] + np.log(row[1]), m + np.log(s1), rtol=1e-13)
        np.testing.assert_allclose(row[1] ** 2 / row[2], s1 ** 2 / s2, rtol=1e-12)
    # a rank with no finite rows must not poison the merge
    empty = torch.tensor([[-np.inf, 0.0, 0.0]], dtype=torch.float64)
    one = torch.tensor([[1.5, 2.0, 3.0]], dtype=torch.float64)
    mm = comm.merge_triples(empty if rank == 0 else one)
    np.testing.assert_allclose(mm[0], [1.5, 2.0 * (world - 1), 3.0 * (world - 1)])
    # sums
    t = torch.tensor([float(rank + 1), 2.0], dtype=torch.float64)
    comm.all_reduce_sum(t)
    assert t.tolist() == [world * (world + 1) / 2, 2.0 * world]
    assert comm.sum_int(10 + rank) == sum(10 + r for r in range(world))
    # all-to-all-v of rows: rank r sends (r + d + 1) rows tagged (r, d) to rank d
    rows, counts = [], []
    for d in range(world):
        k = rank + d + 1
        counts.append(k)
        rows.append(torch.full((k, 3), float(10 * rank + d), dtype=torch.float64))
    send = torch.cat(rows)
    rc = comm.all_to_all_counts(torch.tensor(counts))
    assert rc.tolist() == [s + rank + 1 for s in range(world)]
    recv = comm.all_to_all_rows(send, counts, rc.tolist())
    want = torch.cat([torch.full((s + rank + 1, 3), float(10 * s + rank), dtype=torch.float64) for s in range(world)])
    assert torch.equal(recv, want)
    comm.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def sharded_gpu_worker(rank, world, port, out_dir):
    """2 ranks sharing cuda:0 over gloo (host-staged collectives): sharded resampling partitions the slots,
    and a sharded Sampler run reproduces the analytic evidence."""
    import json
    import numpy as np
    import torch
    import torch.distributed as dist
    _init(rank, world, port)
    import tempest_amd as tp
    from tempest_amd.comm import Comm
    from tempest_amd.device import HipContext
    from tempest_amd.mcmc import PhiloxStream
    from tempest_amd.sharding import resample_sharded
    from tempest_amd.state_manager import StateManager
    dev = torch.device("cuda", 0)
    comm = Comm()
    res = {}
    # ---- sharded resampling: every slot gets exactly one row; frequencies follow the global weights
    for scheme in ("mult", "syst"):
        d, nh, n_local = 3, 5000, 4000
        rs = np.random.RandomState(100 + rank)
        st = StateManager(d, device=0, comm=comm)
        u = rs.rand(nh, d)
        tag = np.full(nh, float(rank))                      # logl carries (rank, row) so rows can be traced
        st.ctx.history_load(u, 20 * u - 10, tag * 1e6 + np.arange(nh), [0.0], [0.0], [nh], [nh * world])
        w = np.exp(rs.randn(nh) * (1.0 + rank))
        tot = torch.tensor([w.sum()], dtype=torch.float64)
        comm.all_reduce_sum(tot)
        wn = w / float(tot)
        ut, xt, lt = resample_sharded(st, torch.from_numpy(wn).to(dev), scheme, PhiloxStream(7), n_local)
        assert ut.shape == (d, n_local) and xt.shape == (d, n_local) and lt.shape == (n_local,)
        ids = lt.cpu().numpy()
        src_rank = (ids // 1e6).astype(int)
        src_row = (ids % 1e6).astype(int)
        np.testing.assert_allclose(xt.cpu().numpy(), 20 * ut.cpu().numpy() - 10, rtol=1e-15)
        # rows really are the source rank's rows
        for r in range(world):
            ur = np.random.RandomState(100 + r).rand(nh, d)
            sel = src_rank == r
            np.testing.assert_array_equal(ut.cpu().numpy().T[sel], ur[src_row[sel]])
        res[scheme] = {"from0": int((src_rank == 0).sum()), "w0": None}
        # share of slots drawn from rank 0 ~ its share of the global weight
        w_all = comm.all_gather(torch.tensor([wn.sum()], dtype=torch.float64)).reshape(-1).numpy()
        cnt = torch.tensor([float((src_rank == 0).sum())], dtype=torch.float64)
        comm.all_reduce_sum(cnt)
        frac0 = float(cnt) / (n_local * world)
        assert abs(frac0 - w_all[0]) < 0.03, (scheme, frac0, w_all)
        res[scheme]["w0"] = float(w_all[0]); res[scheme]["frac0"] = frac0
    # ---- a sharded run end to end
    mean = torch.linspace(-2, 2, 6, dtype=torch.float64, device=dev)

    def loglike(x):
        return -0.5 * ((x - mean) ** 2).sum(dim=1) - 3.0 * float(np.log(2 * np.pi))
    logzs = []
    for seed in range(3):
        s = tp.Sampler(lambda uu: 20 * uu - 10, loglike, 6, n_particles=512, vectorize=True, clustering=False,
                       random_state=seed, device=0)
        assert s.state.comm is not None and s._core.n_local == 256
        s.run(n_total=2048, progress=False)
        logzs.append(s.evidence()[0])
        assert s.state.ctx.size == 256 * len(s.state.get_history("beta"))
    x, w, logl = s.posterior()
    nh_global = 512 * len(s.state.get_history("beta"))
    assert abs(w.sum() - 1.0) < 1e-10 and x.shape[1] == 6 and len(w) <= nh_global and len(w) > nh_global // 4
    xa, wa, _ = s.posterior(trim_importance_weights=False)
    assert len(wa) == nh_global                                             # rows of BOTH shards on every rank
    np.testing.assert_allclose(np.average(x, weights=w, axis=0), np.linspace(-2, 2, 6), atol=0.15)
    # ---- sharded checkpoint: one directory, one raw shard per rank, resume continues identically on every rank
    mk = lambda: tp.Sampler(lambda uu: 20 * uu - 10, loglike, 6, n_particles=512, vectorize=True, clustering=False,  # noqa: E731
                            random_state=11, device=0, output_dir=os.path.join(out_dir, "ck"), output_label="ps")
    sa = mk()
    for _ in range(5):
        sa.sample(return_state=False)
    ck = os.path.join(out_dir, "mid.ckpt")
    sa.save_state(ck)
    comm.barrier()
    assert sorted(os.listdir(ck)) == sorted(["meta.json"] + [f"shard{r:04d}.{n}" for r in range(world) for n in (
        "hist_u.f64", "hist_x.f64", "hist_logl.f64", "cur_u.f64", "cur_x.f64", "cur_logl.f64", "cur_assign.i32")])
    meta = json.load(open(os.path.join(ck, "meta.json")))
    assert meta["world_size"] == world and meta["n_local_t"] == [256] * 5 and meta["n_global_t"] == [512] * 5
    sb = mk()
    sb.run(n_total=2048, progress=False, resume_state_path=ck, save_every=3)
    sc = mk()
    sc.run(n_total=2048, progress=False)
    assert sb.evidence()[0] == sc.evidence()[0]
    np.testing.assert_array_equal(sb.state.get_history("logl", flat=True), sc.state.get_history("logl", flat=True))
    comm.barrier()
    names = sorted(os.listdir(os.path.join(out_dir, "ck")))
    assert "ps_final.ckpt" in names and all(n.endswith(".ckpt") for n in names), names
    res["logz"] = logzs
    res["analytic"] = float(-6 * np.log(20.0))
    assert all(abs(z - res["analytic"]) < 0.35 for z in logzs), logzs
    zt = torch.tensor(logzs, dtype=torch.float64)
    z0 = zt.clone()
    comm.all_reduce_sum(zt)
    assert torch.allclose(zt, world * z0, rtol=0, atol=1e-12)            # every rank holds the same evidence
    json.dump(res, open(os.path.join(out_dir, f"res{rank}.json"), "w"))
    comm.barrier()
    dist.destroy_process_group()
    _ = HipContext


# ---------------------------------------------------------------------------------------------------------------------
# world-size parity: a sharded run is the SAME sampler as the one-GPU run (same seed -> same schedule, same evidence)
PARITY_CASES = {
    "tpcn_mult": dict(sample="tpcn", resample="mult", clustering=False, n_dim=6, target="gauss", n=512, n_total=2048),
    "rwm_syst": dict(sample="rwm", resample="syst", clustering=False, n_dim=6, target="gauss", n=512, n_total=2048),
    "tpcn_mult_bc": dict(sample="tpcn", resample="mult", clustering=False, n_dim=4, target="gauss", n=512, n_total=2048,
                         periodic=[0], reflective=[2]),
    "tpcn_cluster": dict(sample="tpcn", resample="mult", clustering=True, n_dim=4, target="rosen", n=1024, n_total=4096),
    "rwm_cluster_thin": dict(sample="rwm", resample="syst", clustering=True, n_dim=4, target="bimodal", n=1024, n_total=4096,
                             max_points=3000),
}


# d > 16: the proposal kernel of a step follows the redraw probe of the RANK's own shard (blocked kernel in rounds / row walker,
# lane groups and rounds sized by the shard), and those kernels agree to rounding, not bit for bit: a sharded run is the same
# sampler statistically -- same schedule within an iteration, evidence within the seed-to-seed spread
LOOSE_CASES = {
    "tpcn_d24": dict(sample="tpcn", resample="mult", clustering=False, n_dim=24, target="gauss", n=2048, n_total=4096),
    "rwm_d24": dict(sample="rwm", resample="syst", clustering=False, n_dim=24, target="gauss", n=2048, n_total=4096),
}


def parity_run(name, device=0):
    """One seeded run of PARITY_CASES[name] on the current process group (or none): the quantities every world size
    must agree on."""
    import numpy as np
    import torch
    import tempest_amd as tp
    c = PARITY_CASES[name] if name in PARITY_CASES else LOOSE_CASES[name]
    d = c["n_dim"]
    dev = torch.device("cuda", device)
    mean = torch.linspace(-2, 2, d, dtype=torch.float64, device=dev)

    def gauss(x):
        return -0.5 * ((x - mean) ** 2).sum(dim=1) - 0.5 * d * float(np.log(2 * np.pi))

    def rosen(x):
        return -(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2).sum(dim=1)

    def bimodal(x):
        a = -0.5 * (((x - 3.0) / 0.5) ** 2).sum(dim=1)
        b = -0.5 * (((x + 3.0) / 0.5) ** 2).sum(dim=1)
        return torch.logaddexp(a, b)
    like = {"gauss": gauss, "rosen": rosen, "bimodal": bimodal}[c["target"]]
    s = tp.Sampler(lambda u: 20 * u - 10, like, d, n_particles=c["n"], vectorize=True, clustering=c["clustering"],
                   sample=c["sample"], resample=c["resample"], random_state=5, device=device,
                   periodic=c.get("periodic"), reflective=c.get("reflective"))
    if c.get("max_points") and s._core.trainer.clusterer is not None:
        s._core.trainer.clusterer.max_points = c["max_points"]
    ks = []
    core = s._core
    orig = core.trainer.run

    def spy(weights):
        ms = orig(weights)
        ks.append(int(ms.K))
        return ms
    core.trainer.run = spy
    s.run(n_total=c["n_total"], progress=False)
    st = s.state
    x, w, _ = s.posterior()
    return {"logz": float(s.evidence()[0]), "beta": [float(v) for v in st.get_history("beta")],
            "steps": [int(v) for v in st.get_history("steps")], "logz_t": [float(v) for v in st.get_history("logz")],
            "ess": [float(v) for v in st.get_history("ess")], "acc": [float(v) for v in st.get_history("acceptance")],
            "K": ks, "post_n": int(len(w)), "post_mean": [float(v) for v in np.average(x, weights=w, axis=0)],
            "post_wsum": float(w.sum())}


def p2p_gpu_worker(rank, world, port, out_dir):
    """`world` processes sharing cuda:0: the library's peer-to-peer small-message collectives (HIP IPC-mapped inboxes,
    tph_comm_p2p_*) against the values every rank can compute by itself; sizes on both sides of the 32 KB slot."""
    import json
    import numpy as np
    import torch
    import torch.distributed as dist
    from tempest_amd.comm import Comm
    from tempest_amd.device import HipContext
    _init(rank, world, port)
    comm = Comm()
    ctx = HipContext(3, 0)
    comm.attach(ctx, nbytes=4 << 20)
    res = {"p2p": bool(ctx.p2p_active)}
    assert res["p2p"], "peer-to-peer collectives did not attach"
    dev = ctx.device
    tri = world * (world + 1) // 2
    for rep in range(40):                          # the ring of two slots per source is reused 20 times
        for n in (1, 3, 257, 4096, 4097, 20000):   # 4097 doubles and more go through the callback
            base = torch.arange(n, dtype=torch.float64, device=dev) + rep
            t = base * (rank + 1)
            comm.all_reduce_sum(t)
            assert torch.equal(t, base * tri), (rep, n)
        ti = torch.arange(100, dtype=torch.int64, device=dev) * (rank + 1)
        ctx.allreduce_dev(ti, 0)
        assert torch.equal(ti, torch.arange(100, dtype=torch.int64, device=dev) * tri)
        tm = torch.full((33,), rank + rep, dtype=torch.int32, device=dev)
        lo = tm.clone()
        ctx.allreduce_dev(tm, 1)
        ctx.allreduce_dev(lo, 2)
        assert int(tm[0]) == world - 1 + rep and int(lo[7]) == rep
    # the library's own collectives: global reweight triples of a sharded history == one pass over the whole
    rs = np.random.RandomState(1)
    n = 4096
    u = rs.rand(3, n)
    logl = -5.0 * rs.chisquare(3, size=n)
    sl = slice(rank * n // world, (rank + 1) * n // world)
    ctx.history_append(torch.from_numpy(u[:, sl].copy()).to(dev), torch.from_numpy(u[:, sl].copy()).to(dev),
                       torch.from_numpy(logl[sl].copy()).to(dev), 0.0, 0.0, n_global=n)
    got = ctx.reweight_eval([0.0, 0.3, 1.0])
    from oracle import ps
    cm = ps.log_mixture(logl, [0.0], [0.0], [n])
    for b, row in zip([0.0, 0.3, 1.0], got):
        m, s1, s2 = ps.reweight_triple(logl, cm, b)
        np.testing.assert_allclose(row[0] + np.log(row[1]), m + np.log(s1), rtol=1e-12)
        np.testing.assert_allclose(row[1] ** 2 / row[2], s1 ** 2 / s2, rtol=1e-11)
    res["triples"] = [[float(v) for v in row] for row in got]
    # one-sided resample shuffle: slot k is held by rank (7 k) % world as its local row (13 k) % n_hist; every rank can
    # compute what each of its slots must receive.  Three rounds: window reuse, then a larger n_local (window regrowth).
    ctx2 = HipContext(3, 0)
    comm2 = Comm()
    comm2.attach(ctx2, nbytes=4 << 20)
    n_hist = 1000
    def rows_of(holder):
        r = torch.arange(n_hist, dtype=torch.float64, device=dev)
        return torch.stack([holder * 1e6 + r * 100 + c for c in range(3)])
    mine = rows_of(rank)
    ctx2.history_append(mine, mine + 0.5, -mine[0], 0.0, 0.0, n_global=n_hist * world)
    for n_local in (640, 640, 5000):
        k = torch.arange(n_local * world, dtype=torch.int64, device=dev)
        holder, row = (7 * k) % world, (13 * k) % n_hist
        idx = torch.where(holder == rank, row, torch.full_like(row, -1))
        u, x, logl = ctx2.resample_put_global(idx, n_local)
        sl = slice(rank * n_local, (rank + 1) * n_local)
        want = torch.stack([holder[sl] * 1e6 + row[sl] * 100.0 + c for c in range(3)]).to(torch.float64)
        assert torch.equal(u, want) and torch.equal(x, want + 0.5) and torch.equal(logl, -want[0]), n_local
    torch.cuda.synchronize()
    ctx2.p2p_status()
    ctx.p2p_status()
    ctx2.close()
    # a rank that cannot provide its window (forced on rank 0): every rank gets "unavailable" and nothing hangs
    os.environ["TEMPEST_AMD_P2P_NOWINDOW"] = "1"
    ctx3 = HipContext(3, 0)
    comm3 = Comm()
    comm3.attach(ctx3, nbytes=4 << 20)
    ctx3.history_append(mine, mine + 0.5, -mine[0], 0.0, 0.0, n_global=n_hist * world)
    k = torch.arange(64 * world, dtype=torch.int64, device=dev)
    idx = torch.where((7 * k) % world == rank, (13 * k) % n_hist, torch.full_like(k, -1))
    assert ctx3.resample_put_global(idx, 64) is None and ctx3.resample_put_global(idx, 64) is None
    t = torch.ones(3, dtype=torch.float64, device=dev)
    comm3.all_reduce_sum(t)                          # the small collectives are unaffected
    assert t.tolist() == [float(world)] * 3
    del os.environ["TEMPEST_AMD_P2P_NOWINDOW"]
    res["fallback"] = True
    ctx3.close()
    # a peer that never arrives: the waiting kernel gives up after the timeout, the error is reported (not a hang), and every
    # later exchange of that context fails at once
    import time
    from tempest_amd._lib import TempestHipError
    keep = os.environ.get("TEMPEST_AMD_P2P_TIMEOUT")
    os.environ["TEMPEST_AMD_P2P_TIMEOUT"] = "2"
    ctx4 = HipContext(3, 0)
    comm4 = Comm()
    comm4.attach(ctx4, nbytes=4 << 20)
    assert ctx4.p2p_active
    dist.barrier()
    if rank == 0:
        t = torch.ones(4, dtype=torch.float64, device=dev)
        t0 = time.perf_counter()
        ctx4.allreduce_dev(t, 0)                     # nobody else takes part
        torch.cuda.synchronize()
        assert 1.5 < time.perf_counter() - t0 < 30.0
        try:
            ctx4.p2p_status()
            raise AssertionError("the timed-out exchange was not reported")
        except TempestHipError as e:
            assert "timed out" in str(e)
        t0 = time.perf_counter()
        try:
            ctx4.allreduce_dev(t, 0)
            raise AssertionError("an exchange after the failure was accepted")
        except TempestHipError:
            pass
        assert time.perf_counter() - t0 < 1.0
    dist.barrier()
    if keep is None:
        del os.environ["TEMPEST_AMD_P2P_TIMEOUT"]
    else:
        os.environ["TEMPEST_AMD_P2P_TIMEOUT"] = keep
    res["timeout_reported"] = True
    ctx4.close()
    json.dump(res, open(os.path.join(out_dir, f"p2p{rank}.json"), "w"))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def parity_gpu_worker(rank, world, port, out_dir):
    """`world` ranks sharing cuda:0 over gloo: every case of PARITY_CASES through the sharded path (small collectives through
    the library's peer-to-peer exchange; TEMPEST_AMD_P2P=0 in the environment: through the process group)."""
    import json
    import torch.distributed as dist
    _init(rank, world, port)
    only = os.environ.get("TEMPEST_AMD_TEST_CASES")
    cases = LOOSE_CASES if os.environ.get("TEMPEST_AMD_TEST_LOOSE") == "1" else PARITY_CASES
    out = {name: parity_run(name) for name in cases if not only or name in only.split(",")}
    json.dump(out, open(os.path.join(out_dir, f"parity{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# BITWISE world-size invariance: with the canonical partition (csrc/common.h: tph_part) every global
# quantity -- reweight triples, cumulative weights, the moments of the proposal fit, the acceptance sums of every MCMC step --
# is formed per virtual shard and folded in shard order, so a run on G ranks is the SAME floating-point computation as the run
# on one, for every G that divides V(n_particles).  49 152 = 3 * 2^14 particles: V = 48 (worlds 1, 2, 3, 4, ...); 65 536: V = 16
# (worlds 1, 2, 4, ...: the size at which a two-rank run used to part from the one-rank run at iteration 17).  Above 16
# dimensions the run is pinned to the screened batches (TEMPEST_AMD_REGIME=screened: the adaptive regime picks kernels by each
# rank's own list lengths, and those kernels agree to rounding only).
BITWISE_CASES = {
    "rosen10_49152": dict(n_dim=10, target="rosen", n=49152, worlds=(2, 3, 4), sample="tpcn", resample="mult"),
    "rosen10_65536": dict(n_dim=10, target="rosen", n=65536, worlds=(2, 4), sample="tpcn", resample="mult"),
    "gauss10_syst_12288": dict(n_dim=10, target="gauss", n=12288, worlds=(2, 3, 4), sample="rwm", resample="syst"),
    "gauss50_12288": dict(n_dim=50, target="gauss", n=12288, worlds=(2, 3, 4), sample="tpcn", resample="mult", pin="screened"),
    # dynamic mode (tempest/steps/reweight.py:427-495): the beta search is driven by the volume variation of the weighted history,
    # whose moments and Mahalanobis sum are formed over the same partition
    "gauss10_dynamic_12288": dict(n_dim=10, target="gauss", n=12288, worlds=(2, 3, 4), sample="tpcn", resample="mult", vv=0.5),
}


def bitwise_run(name, device=0):
    """One seeded run of BITWISE_CASES[name] on the current process group (or none).  Everything returned must be EQUAL for every
    world size: the schedule, every iteration's evidence / ESS / acceptance, the final evidence and a digest of the final ensemble
    (u and logl of all ranks in slot order) and of the whole history's log-likelihoods."""
    import hashlib
    import numpy as np
    import torch
    import tempest_amd as tp
    c = BITWISE_CASES[name]
    d = c["n_dim"]
    dev = torch.device("cuda", device)
    mean = torch.linspace(-2, 2, d, dtype=torch.float64, device=dev)

    def gauss(x):
        return -0.5 * ((x - mean) ** 2).sum(dim=1) - 0.5 * d * float(np.log(2 * np.pi))

    def rosen(x):
        return -(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2).sum(dim=1)
    keep = os.environ.get("TEMPEST_AMD_REGIME")
    if c.get("pin"):
        os.environ["TEMPEST_AMD_REGIME"] = c["pin"]
    try:
        s = tp.Sampler(lambda u: 20 * u - 10, {"gauss": gauss, "rosen": rosen}[c["target"]], d, n_particles=c["n"], vectorize=True,
                       clustering=False, sample=c["sample"], resample=c["resample"], random_state=5, device=device,
                       volume_variation=c.get("vv"))
        s.run(n_total=2 * c["n"], progress=False)
    finally:
        if c.get("pin"):
            if keep is None:
                del os.environ["TEMPEST_AMD_REGIME"]
            else:
                os.environ["TEMPEST_AMD_REGIME"] = keep
    st = s.state
    comm = st.comm
    u = st.get_current("u")
    logl = st.get_current("logl")
    hist = np.asarray(st.get_history("logl", flat=True))
    if comm is not None and comm.active:
        u, logl = comm.gather_rows(u), comm.gather_rows(logl)
        # the history in GLOBAL order: iteration-major, then rank (each rank holds its slots of every iteration)
        T = len(st.get_history("beta"))
        per = hist.size // T
        allh = comm.gather_rows(hist.reshape(T, per).T.copy())            # (world * per, T), rank-major rows
        hist = allh.T.reshape(-1)                                         # iteration-major; inside: rank, then slot = global slot order
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(u).tobytes())
    h.update(np.ascontiguousarray(logl).tobytes())
    hh = hashlib.sha256(np.ascontiguousarray(hist).tobytes())
    return {"logz": float(s.evidence()[0]).hex(), "beta": [float(v).hex() for v in st.get_history("beta")],
            "steps": [int(v) for v in st.get_history("steps")], "logz_t": [float(v).hex() for v in st.get_history("logz")],
            "ess": [float(v).hex() for v in st.get_history("ess")], "acc": [float(v).hex() for v in st.get_history("acceptance")],
            "eff": [float(v).hex() for v in st.get_history("efficiency")], "cv": [float(v).hex() for v in st.get_history("cv")],
            "ensemble_sha256": h.hexdigest(),
            "history_logl_sha256": hh.hexdigest(), "logz_float": float(s.evidence()[0])}


def bitwise_gpu_worker(rank, world, port, out_dir):
    """`world` ranks sharing cuda:0 over gloo: every case of BITWISE_CASES that admits this world size."""
    import json
    import torch.distributed as dist
    _init(rank, world, port)
    only = os.environ.get("TEMPEST_AMD_TEST_CASES")
    out = {name: bitwise_run(name) for name, c in BITWISE_CASES.items()
           if world in c["worlds"] and (not only or name in only.split(","))}
    json.dump(out, open(os.path.join(out_dir, f"bitwise{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()
