"""The MCMC step as a replayable hipGraph (tempest_amd/mcmc.py: StepEngine): device-side step control
(tick offset, beta, stop flag), the pinned-host mailbox of tph_adapt, and whole runs that must be bit-identical to the
step-by-step launch path."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def prior20(u):
    return 20 * u - 10


def rosen(x):
    return -(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2).sum(dim=1)


def bimodal(x):
    a = -0.5 * (((x - 4.0) / 0.5) ** 2).sum(dim=1)
    b = -0.5 * (((x + 4.0) / 0.5) ** 2).sum(dim=1)
    return torch.logaddexp(a, b)


def _history(s):
    st = s.state
    return {k: np.asarray(st.get_history(k)) for k in ("beta", "logz", "steps", "acceptance", "efficiency")}


@pytest.mark.parametrize("kernel,clustering,like,d", [("tpcn", False, rosen, 4), ("rwm", False, rosen, 4),
                                                      ("tpcn", True, bimodal, 3)])
def test_graph_run_is_bit_identical(kernel, clustering, like, d):
    import tempest_amd as tp
    runs = []
    for graph in (False, True):
        s = tp.Sampler(prior20, like, d, n_particles=512, vectorize=True, clustering=clustering, random_state=5,
                       sample=kernel, graph=graph, periodic=[0] if kernel == "rwm" else None)
        s.run(n_total=2048, progress=False)
        runs.append((s.evidence()[0], _history(s), s.posterior()[0], s))
    eng = runs[1][3]._core.mutator._engines
    assert eng and any(e.graph is not None for e in eng.values()), "the step was never replayed as a graph"
    assert all(e.graph is None for e in runs[0][3]._core.mutator._engines.values())
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k], err_msg=k)
    np.testing.assert_array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("clustering", [False, True])
def test_graph_run_above_16_dimensions_makes_the_same_run(clustering, monkeypatch):
    """The same at n_dim = 24, where a step is screened batches first and matrix-core rounds (fanned-out list rounds, the screened
    or the multi-lane straggler pass) later -- with one mode, and with clustering on a separable four-mode target that the split
    search does split at this size (several modes: mode-pure tiles, per-mode lists and attempts): graph replay and step-by-step
    launches make the same run."""
    import tempest_amd as tp
    from tempest_amd.steps import train as tr
    d = 24
    dev = torch.device("cuda", 0)
    mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
    for k, (a, b) in enumerate([(-6, -6), (-6, 6), (6, -6), (6, 6)]):
        mus[k, 0], mus[k, 1] = a, b

    def like(x):
        q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
        return torch.logsumexp(-0.5 * q / 0.09, dim=1)
    ks = []
    orig = tr.Trainer.run

    def trun(self, w):
        ms = orig(self, w)
        ks.append(int(ms.K))
        return ms
    monkeypatch.setattr(tr.Trainer, "run", trun)
    from tempest_amd import mcmc
    captured = []                                        # (K of the engine) at every capture that produced a graph
    orig_capture = mcmc.StepEngine._capture

    def capture(self):
        orig_capture(self)
        if self.graph is not None:
            captured.append(self.K)
    monkeypatch.setattr(mcmc.StepEngine, "_capture", capture)
    runs = []
    for graph in (False, True, True):
        ks.clear()
        s = tp.Sampler(prior20, like, d, n_particles=1024, vectorize=True, clustering=clustering, random_state=11,
                       sample="tpcn", graph=graph)
        s.run(n_total=2048, progress=False)
        runs.append((s.evidence()[0], _history(s), s.posterior()[0], s, list(ks)))
    assert captured, "the step was never replayed as a graph"
    if clustering:
        assert any(K >= 2 for K in captured), captured      # an engine with several modes was captured and replayed
    if clustering:
        assert max(runs[0][4]) >= 2, runs[0][4]            # several modes were in play
    # (Not bit for bit, unlike d <= 16: a captured graph keeps the number of rounds it was captured with while the eager path
    # follows the regime rule step by step, so some winning attempts are evaluated by a matrix-core round in one run and by the
    # FP64 chain of the screened kernel in the other -- the same attempt, the same proposal to the last bit or two.  The runs make
    # the same decisions: same cluster counts, same steps per iteration, evidence and posterior equal to rounding.)
    assert runs[0][4] == runs[1][4]
    np.testing.assert_array_equal(runs[0][1]["steps"], runs[1][1]["steps"])
    np.testing.assert_array_equal(runs[0][1]["beta"].shape, runs[1][1]["beta"].shape)
    for k in ("beta", "logz", "acceptance", "efficiency"):
        np.testing.assert_allclose(runs[0][1][k], runs[1][1][k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(runs[0][0], runs[1][0], rtol=1e-10)
    np.testing.assert_allclose(runs[0][2], runs[1][2], rtol=1e-8, atol=1e-10)
    # two runs of the same kind: bitwise the same (the order in which failures enter the lists of a round varies from launch to
    # launch, a particle's attempts and arithmetic do not depend on it)
    assert runs[1][0] == runs[2][0]
    np.testing.assert_array_equal(runs[1][2], runs[2][2])
    for k in runs[1][1]:
        np.testing.assert_array_equal(runs[1][1][k], runs[2][1][k], err_msg=k)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_graph_replay_of_the_blocked_proposal_follows_new_mode_statistics(kernel):
    """d > 16, every dimension reflective: every first attempt is in bounds, so the engine switches to the blocked proposal
    kernel during its first run and captures the step with it in the second.  The row-blocked copies of L and L^-1 that kernel
    reads are rebuilt by host code a replayed graph never re-enters: each PS iteration loads new mode statistics, and the
    replayed step must see them (the rebuild is recorded into the graph) -- whole runs bit-identical to step-by-step launches."""
    import tempest_amd as tp
    d = 20
    mean = torch.linspace(-3, 3, d, dtype=torch.float64, device="cuda:0")

    def like(x):
        return -0.5 * (((x - mean) / 0.7) ** 2).sum(dim=1)
    runs = []
    for graph in (False, True):
        s = tp.Sampler(prior20, like, d, n_particles=512, vectorize=True, clustering=False, random_state=11,
                       sample=kernel, graph=graph, reflective=list(range(d)))
        s.run(n_total=2048, progress=False)
        runs.append((s.evidence()[0], _history(s), s.posterior()[0], s))
    eng = list(runs[1][3]._core.mutator._engines.values())
    assert eng and any(e.graph is not None and e.blocked and e.runs >= 4 for e in eng), \
        "the blocked proposal kernel was never replayed as a graph over several iterations"
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k], err_msg=k)
    np.testing.assert_array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
@pytest.mark.parametrize("variant,rounds", [(5, 0), (4, 0), (4, 4), (6, 0), (3, 0)])
def test_d_gt_16_proposal_paths_replayed_from_a_graph_equal_eager_launches(kernel, variant, rounds):
    """tph_propose at n_dim > 16 captured ONCE (row-walker kernel 5, blocked kernel + straggler pass 4 -- also in four rounds, whose
    list rounds fan out by a width and from an attempt that live in device memory --, screened batches 6, multi-lane kernel 3) and
    replayed after the caller has rewritten its fixed-address inputs -- new mode statistics, new positions, new step-control
    block: every replay must equal the eager launch on the same inputs bit for bit.  Pins the two things a replay cannot get
    from the host: the library's derived copies of L / L^-1 (rebuilt inside the graph) and its per-launch work-queue words
    (zeroed by a kernel: a captured hipMemsetAsync node was seen to run out of order on the second replay)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from types import SimpleNamespace
    from oracle import ps
    from tempest_amd.device import HipContext
    dev = torch.device("cuda", 0)
    d, n = 32, 1000
    rs = np.random.RandomState(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731

    def mk_modes(scale):
        A = rs.randn(d, d) / np.sqrt(d)
        cov = ((A @ A.T + np.eye(d)) * (scale ** 2 / 2.0))[None]
        means = 0.5 + 0.02 * rs.randn(1, d)
        _, chol, _ = ps.mode_statistics(means, cov)
        return means, chol, np.linalg.inv(chol[0])[None]
    c = HipContext(d, device=0)
    c.set_option(0, variant)
    c.set_option(10, 1)
    c.set_option(4, rounds)
    m0 = mk_modes(0.29)
    modes = SimpleNamespace(K=1, means_dev=t(m0[0]), chol_dev=t(m0[1]), winv_dev=t(m0[2]), dof_dev=t(np.array([1e6])))
    u, up, mu_, mup = t(rs.rand(d, n)), c.empty(d, n), c.empty(n), c.empty(n)
    sig = t(np.array([2.38 / np.sqrt(d) if kernel == "rwm" else 0.6]))
    ctl = c.zeros(10)
    pend = torch.zeros(n, dtype=torch.uint8, device=dev)

    def call():
        c.propose(kernel, u, None, modes, sig, None, 77, 1, 0, up, mu_, mup, ctl=ctl, pending=pend)
    call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        c.use_current_stream()
        g.capture_begin()
        call()
        g.capture_end()
    c.use_current_stream()
    torch.cuda.current_stream(dev).wait_stream(side)
    for trial in range(4):
        m = mk_modes(0.29 if trial < 2 else 0.12)             # trials 2, 3: most first attempts in bounds
        modes.means_dev.copy_(t(m[0])); modes.chol_dev.copy_(t(m[1])); modes.winv_dev.copy_(t(m[2]))
        u.copy_(t(np.clip(0.5 + (rs.rand(d, n) - 0.5) * (1.0 if trial < 2 else 0.5), 0.0, 1.0)))
        ctl[0], ctl[7] = float(trial), 10.0 * trial           # steps done (the form at u is carried from trial 1 on), tick base
        c.set_option(5, 100 + trial)                          # TPH_OPT_MODES_EPOCH: what StepEngine.load() does
        keep = mu_.clone()
        g.replay()
        torch.cuda.synchronize()
        a = (up.clone(), mu_.clone(), mup.clone(), float(ctl[8]))
        mu_.copy_(keep)
        call()
        torch.cuda.synchronize()
        b = (up.clone(), mu_.clone(), mup.clone(), float(ctl[8]))
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y), (trial, variant)
        if variant != 4 or trial < 2:
            assert a[3] == b[3] and a[3] >= 1.0
    c.close()


def test_uncapturable_callback_falls_back():
    """A likelihood that synchronises with the host cannot be stream-captured: the engine warns once and keeps launching
    step by step, with the same numbers."""
    import tempest_amd as tp

    def like_sync(x):
        ll = -0.5 * (x ** 2).sum(dim=1)
        if float(ll.max().item()) > 1e300:       # host round trip
            raise RuntimeError
        return ll

    def like(x):
        return -0.5 * (x ** 2).sum(dim=1)
    s0 = tp.Sampler(prior20, like, 3, n_particles=256, vectorize=True, clustering=False, random_state=2, graph=False)
    s0.run(n_total=1024, progress=False)
    s1 = tp.Sampler(prior20, like_sync, 3, n_particles=256, vectorize=True, clustering=False, random_state=2, graph=True)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        s1.run(n_total=1024, progress=False)
    assert any("could not be captured" in str(w.message) for w in rec)
    eng = list(s1._core.mutator._engines.values())
    assert eng and all(e.graph is None and e.graph_error is not None for e in eng)
    assert s0.evidence()[0] == s1.evidence()[0]
    # the device is still usable afterwards
    s2 = tp.Sampler(prior20, like, 3, n_particles=256, vectorize=True, clustering=False, random_state=2, graph=True)
    s2.run(n_total=1024, progress=False)
    assert s2.evidence()[0] == s0.evidence()[0]


def test_step_control_block_semantics():
    """tph_propose / tph_accept / tph_adapt with a device-resident control block: tick = tick + ctl[7] + 2 ctl[0], beta from
    ctl[6], no-ops once ctl[1] is set; tph_adapt's record arrives in the pinned mailbox."""
    from types import SimpleNamespace
    from tempest_amd.device import HipContext
    d, n, K = 5, 1000, 1
    dev = torch.device("cuda", 0)
    ctx = HipContext(d, device=0)
    g = torch.Generator(device="cpu").manual_seed(1)
    u = (0.3 + 0.4 * torch.rand(d, n, generator=g, dtype=torch.float64)).to(dev)
    A = torch.randn(d, d, generator=g, dtype=torch.float64)
    cov = (A @ A.T / d + torch.eye(d, dtype=torch.float64)) * 1e-3
    modes = SimpleNamespace(K=1, means_dev=torch.full((1, d), 0.5, dtype=torch.float64, device=dev),
                            chol_dev=torch.linalg.cholesky(cov).reshape(1, d, d).contiguous().to(dev),
                            inv_dev=torch.linalg.inv(cov).reshape(1, d, d).contiguous().to(dev),
                            dof_dev=torch.full((1,), 1e6, dtype=torch.float64, device=dev))
    sig = torch.full((1,), 0.5, dtype=torch.float64, device=dev)
    seed, base, done_steps = 1234, 100, 3

    def propose(tick, ctl, maha_u=None):
        up, mu, mup = ctx.empty(d, n), (ctx.empty(n) if maha_u is None else maha_u.clone()), ctx.empty(n)
        ctx.propose("tpcn", u, None, modes, sig, None, seed, tick, 0, up, mu, mup, ctl=ctl)
        return up, mu, mup
    ctl = torch.tensor([done_steps, 0, 0, 0, 0, 0, 0.37, base, 0, 0], dtype=torch.float64, device=dev)   # TPH_STEP_STATE_LEN
    b = propose(1 + base + 2 * done_steps, None)
    # with steps already done the kernel READS the Mahalanobis form at u from maha_u (kept current by tph_accept)
    a = propose(1, ctl, maha_u=b[1])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # ... and a wrong carried value changes the proposal: it really is read, not recomputed
    c = propose(1, ctl, maha_u=b[1] * 4.0 + 1.0)
    assert not torch.equal(c[0], b[0])
    # at step 0 of a run it is computed whatever the buffer holds
    ctl0 = ctl.clone(); ctl0[0] = 0.0
    e = propose(1, ctl0, maha_u=b[1] * 4.0 + 1.0)
    f = propose(1 + base, None)
    for x, y in zip(e, f):
        assert torch.equal(x, y)
    # accept: beta and tick from the block == by-value call
    up, mu, mup = a
    lp = -0.5 * ((up - 0.5) ** 2).sum(dim=0) * 50
    logl0 = -0.5 * ((u - 0.5) ** 2).sum(dim=0) * 50

    def accept(tick, beta, ctl, partials=None):
        uu, xx, ll, sums = u.clone(), u.clone(), logl0.clone(), ctx.zeros(2)
        ctx.accept("tpcn", beta, uu, xx, ll, up, up, lp, mu, mup, None, K, modes.dof_dev, seed, tick, 0, sums, ctl=ctl,
                   partials=partials)
        return uu, xx, ll, sums
    part = ctx.empty(((n + 255) // 256) * 2)
    mu_before = mu.clone()
    r1 = accept(2, 0.0, ctl, part)
    moved = (r1[2] != logl0)
    assert torch.equal(mu[moved], mup[moved]) and torch.equal(mu[~moved], mu_before[~moved])   # maha_u follows the accepted rows
    mu.copy_(mu_before)
    r2 = accept(2 + base + 2 * done_steps, 0.37, None)
    mu.copy_(mu_before)
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)
    assert 0 < r1[3][0].item() < n
    # stop flag set: nothing moves
    ctl_done = ctl.clone(); ctl_done[1] = 1.0
    r3 = accept(2, 0.0, ctl_done, part)
    assert torch.equal(r3[0], u) and torch.equal(r3[2], logl0) and float(r3[3].abs().sum()) == 0.0
    # adapt: record in the mailbox; a second call after `done` is a no-op
    counts = torch.full((1,), float(n), dtype=torch.float64, device=dev)
    sums = r1[3].clone()
    mailbox = torch.full((4, 8), -1.0, dtype=torch.float64).pin_memory()
    state = torch.zeros(10, dtype=torch.float64, device=dev)
    sg = sig.clone()
    ctx.adapt("tpcn", sums, counts, K, n, 1, 1, sg, state, mailbox=mailbox)    # n_max = n_steps = 1: done after d steps
    ctx.synchronize()
    st = state.cpu().numpy()
    assert st[0] == 1.0 and mailbox[1, 7] == 1.0 and np.array_equal(mailbox[1, :6].numpy(), st[:6])
    for _ in range(d - 1):
        ctx.adapt("tpcn", sums, counts, K, n, 1, 1, sg, state, mailbox=mailbox)
    ctx.synchronize()
    assert state[0].item() == d and state[1].item() == 1.0 and mailbox[d % 4, 7] == d
    frozen, sg_frozen = state.clone(), sg.clone()
    ctx.adapt("tpcn", sums, counts, K, n, 1, 1, sg, state, mailbox=mailbox)
    ctx.synchronize()
    assert torch.equal(state, frozen) and torch.equal(sg, sg_frozen)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_graph_is_captured_again_when_the_redraw_regime_changes_the_proposal_kernel(kernel):
    """d > 16 from the prior: the first iterations are redraw-dominated (row walker), the later ones one attempt per particle
    (blocked kernel).  A graph captured early is retired when the regime rule asks for the other kernel and the step is
    captured again -- the run ends on the blocked kernel, and equals the step-by-step run (same rule, same probe values)."""
    import tempest_amd as tp
    d = 40

    def gauss(x):
        return -0.5 * (x ** 2).sum(dim=1)
    runs = []
    for graph in (False, True):
        s = tp.Sampler(prior20, gauss, d, n_particles=1024, vectorize=True, clustering=False, random_state=3, sample=kernel,
                       graph=graph)
        s.run(n_total=4096, progress=False)
        runs.append((s.evidence()[0], _history(s), s))
    eng = list(runs[1][2]._core.mutator._engines.values())
    assert eng and all(e.graph is not None for e in eng)
    assert any(e._retired_graphs for e in eng), "the regime never changed the kernel under a captured graph"
    assert all(e.blocked >= 1 and not e.staged for e in eng)            # the run ends in the one-attempt regime
    assert abs(runs[0][0] - runs[1][0]) < 1e-9
    np.testing.assert_array_equal(runs[0][1]["steps"], runs[1][1]["steps"])
    np.testing.assert_allclose(runs[0][1]["logz"], runs[1][1]["logz"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("kernel", ["tpcn", "rwm"])
def test_regime_probe_driven_across_its_thresholds_on_the_gpu(kernel, monkeypatch):
    """VERDICT r04 item 8, on the device: the redraw probe of a 40-dimensional run is REPLACED by a script while the steps really
    run -- captured graphs, real kernels on both sides of every switch.  The band of mcmc.REGIME_THRESHOLDS at 40-D: the blocked
    rounds are left at a geometric ESTIMATE of 3.4 attempts, the screened batches at a TRUE mean below 5.5 (the two kernels report
    different probes, the true mean the larger; a script entry holds both).  The script
      (a) sits below both thresholds, (b) rises above both, (c) falls INTO the band (estimate below 3.4, true mean above 5.5: the
      screened batches must stay), (d) falls below both, (c') returns into the band (now the blocked rounds must stay: the same
      probes as in (c), the other kernel), (e) then sits ON the thresholds so that either kernel asks for the other at every
      reading, forty times.
    Asserted: exactly one switch up and one down in (a)-(c'), none inside the band from either side; the forty flips cost a handful
    of switches with growing waits (the dwell rule); every switch under a graph retires exactly that graph and every graph that
    ever existed was captured once; and the run with graphs IS the step-by-step run under the same script: same kernels at every
    reading, same steps, same evidence, bit for bit (the kernels either side of a threshold agree with the oracle to rounding:
    tests/test_screened_gpu.py, test_kernels_gpu.py::test_blocked_*)."""
    import tempest_amd as tp
    from tempest_amd import mcmc
    d = 40
    up_est, down_true, _, _ = mcmc.regime_band("screened", d)
    script = [(1.3, 1.3)] * 6 + [(up_est + 3.0, down_true + 4.0)] * 6 + [(up_est - 0.4, down_true + 0.3)] * 8 + [(1.3, 1.3)] * 6
    script += [(up_est - 0.4, down_true + 0.3)] * 8
    phases_end = len(script)
    script += [(up_est + 0.05, down_true - 0.05)] * 40

    def gauss(x):
        return -0.5 * (x ** 2).sum(dim=1)
    orig_regime, orig_capture = mcmc.StepEngine._regime, mcmc.StepEngine._capture
    runs = []
    for graph in (True, False):
        seen, captures = [], []

        def scripted(self, mean_attempts, seen=seen):
            i = len(seen)
            est, true_mean = script[i] if i < len(script) else (1.3, 1.3)
            orig_regime(self, est if self.blocked > 0 else true_mean)
            seen.append((self.blocked > 0, bool(self.staged), len(self._retired_graphs), self.graph is not None))

        def capture(self, captures=captures):
            captures.append(1)
            return orig_capture(self)
        monkeypatch.setattr(mcmc.StepEngine, "_regime", scripted)
        monkeypatch.setattr(mcmc.StepEngine, "_capture", capture)
        s = tp.Sampler(prior20, gauss, d, n_particles=1024, vectorize=True, clustering=False, random_state=5, sample=kernel,
                       graph=graph)
        s.run(n_total=98304, progress=False)
        assert len(seen) > len(script), ("the run was too short for the script", len(seen))
        runs.append((s.evidence()[0], _history(s), list(seen), len(captures), s))
    monkeypatch.setattr(mcmc.StepEngine, "_regime", orig_regime)
    monkeypatch.setattr(mcmc.StepEngine, "_capture", orig_capture)
    seen = runs[0][2]
    kern = [(b, st) for b, st, _, _ in seen]
    assert kern == [(b, st) for b, st, _, _ in runs[1][2]]                 # the rule does not depend on the launch path
    switches = [i for i in range(1, len(kern)) if kern[i] != kern[i - 1]]
    early = [i for i in switches if i < phases_end]
    assert len(early) == 2, (early, kern[:phases_end])
    assert 6 <= early[0] < 12 and kern[early[0]] == (False, True)           # up: onto the screened batches, in phase (b)
    assert 20 <= early[1] < 26 and kern[early[1]] == (True, False)          # down: only below BOTH thresholds, phase (d)
    assert all(k == (False, True) for k in kern[early[0]:20])               # inside the band nothing moves, from above ...
    assert all(k == (True, False) for k in kern[early[1]:phases_end])       # ... nor from below
    flips = [i for i in switches if phases_end <= i < len(script)]
    assert 1 <= len(flips) <= 8, flips                                      # forty crossings, a handful of switches
    gaps = np.diff(flips)
    assert len(gaps) < 3 or gaps[-1] >= 4                                   # ... and the waits grow
    # graphs: one retired per switch made under a graph, one capture per graph that ever existed
    eng = list(runs[0][4]._core.mutator._engines.values())
    retired = sum(len(e._retired_graphs) for e in eng)
    assert retired >= len(early) and retired <= len(switches)
    assert runs[0][3] == retired + sum(1 for e in eng if e.graph is not None)
    assert runs[1][3] == 0
    # and the samples: the captured run is the step-by-step run
    assert runs[0][0] == runs[1][0]
    for k in ("beta", "logz", "steps", "acceptance"):
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k])
    assert np.isfinite(runs[0][0])
