"""Host-side logic that needs neither a GPU nor the HIP library: checkpoint metadata encoding, format selection, the
detection of a fusable HipCallbacks pair, the size buckets of history-sized temporaries."""
import numpy as np
import pytest


def test_checkpoint_scalar_encoding_roundtrip():
    from tempest_amd import checkpoint as ck
    vals = [None, True, 3, np.int64(7), 2.5, np.float64(-1e300), float("inf"), np.array([1.0, 2.0]), np.array([[1, 2], [3, 4]])]
    for v in vals:
        enc = ck._json_scalar(v)
        dec = ck._from_json_scalar(enc)
        if isinstance(v, np.ndarray):
            np.testing.assert_array_equal(dec, v)
            assert dec.dtype == v.dtype
        else:
            assert dec == v or (v is None and dec is None)
    with pytest.raises(TypeError):
        ck._json_scalar(object())


def test_checkpoint_format_selection(tmp_path):
    from tempest_amd import checkpoint as ck
    assert ck.wants_native(tmp_path / "a.ckpt") and not ck.wants_native(tmp_path / "a.state")
    assert ck.wants_native(tmp_path / "a.state", "native") and not ck.wants_native(tmp_path / "a.ckpt", "dill")
    # a checkpoint caught between the two renames of save() exists as `<name>.old` only: still found
    assert not ck.is_native(tmp_path / "run.ckpt")
    (tmp_path / "run.ckpt.old").mkdir()
    (tmp_path / "run.ckpt.old" / "meta.json").write_text("{}")
    assert ck.is_native(tmp_path / "run.ckpt") and ck._resolve(tmp_path / "run.ckpt").name == "run.ckpt.old"
    (tmp_path / "run.ckpt").mkdir()
    (tmp_path / "run.ckpt" / "meta.json").write_text("{}")
    assert ck._resolve(tmp_path / "run.ckpt").name == "run.ckpt"
    with pytest.raises(ValueError):
        ck.wants_native(tmp_path / "a", "hdf5")
    (tmp_path / "d").mkdir()
    assert ck.wants_native(tmp_path / "d") and not ck.is_native(tmp_path / "d")
    (tmp_path / "d" / "meta.json").write_text("{}")
    assert ck.is_native(tmp_path / "d")

    class FakeComm:
        active = True
    assert ck.wants_native(tmp_path / "x.state", None, FakeComm())       # sharded runs never write one pickle per rank


def test_fused_plugin_detection():
    from tempest_amd.hipcallbacks import HipCallbacks, fused_plugin

    def f(x):
        return x
    assert fused_plugin(f, f) is None
    a = HipCallbacks.__new__(HipCallbacks)
    a.fused, a.whole_step, a.n_dim = True, True, 10
    b = HipCallbacks.__new__(HipCallbacks)
    b.fused, b.whole_step, b.n_dim = True, True, 10
    assert fused_plugin(a.prior_transform, a.log_likelihood) is a
    assert fused_plugin(a.prior_transform, b.log_likelihood) is None     # two different plugins: nothing to fuse
    a.fused = False
    assert fused_plugin(a.prior_transform, a.log_likelihood) is None
    a.fused = True
    assert a.can_fuse_step(1, False, 131072) and not a.can_fuse_step(1, False, 1 << 20)
    assert not a.can_fuse_step(3, True, 1000)
    a.whole_step = "always"
    assert a.can_fuse_step(1, False, 1 << 20)
    a.n_dim = 17
    assert not a.can_fuse_step(1, False, 1000)


def test_history_sized_temporaries_come_from_buckets():
    # the rule of HipContext.empty_rows: capacity = the smallest of {2^k, 1.5 * 2^k} >= n (at least 1024)
    def bucket(n):
        p2 = 1 << max(10, (n - 1).bit_length())
        return p2 * 3 // 4 if p2 * 3 // 4 >= n else p2
    for n in (1, 1024, 1025, 1536, 1537, 2048, 3000, 3073, 10 ** 6, 25 * 10 ** 6):
        c = bucket(n)
        assert c >= n and c <= max(1024, int(1.5 * n) + 1)
        assert c & (c - 1) == 0 or (c // 3) & ((c // 3) - 1) == 0
    import inspect
    from tempest_amd.device import HipContext
    src = inspect.getsource(HipContext.empty_rows)
    assert "bit_length" in src and "3 // 4" in src


def test_proposal_regime_rule_d_gt_16(monkeypatch):
    """(With the screen off, TEMPEST_AMD_SCREEN=0: the FP64 row walker's crossovers; the screened rule: next test.)
    StepEngine._regime (mcmc.py): which d > 16 proposal kernel the NEXT steps get from the redraw probe -- blocked kernel in
    R rounds for a few attempts per particle (R from the expected length of the round's list), row walker when redraws
    dominate, one threshold per direction (the blocked path reports the geometric estimate, the walker the true mean), the
    multi-lane kernel for several modes; a captured graph keeps what it was captured with until the rule asks for another
    kernel."""
    from tempest_amd import mcmc
    from tempest_amd.device import OPT_BLOCKED, OPT_ML_UNSTAGED, OPT_SCREEN, OPT_STAGED_REDRAW
    monkeypatch.setenv("TEMPEST_AMD_SCREEN", "0")

    class Ctx:
        def __init__(self, d):
            self.n_dim, self.opts = d, {}

        def set_option(self, k, v):
            self.opts[k] = v

    class Eng:
        def _regime(self, m):          # the thresholds alone: every reading as if the last switch were long ago (the dwell: last test)
            self._since, self._dwell, self._quick = 1 << 30, 0, 0
            return mcmc.StepEngine._regime(self, m)

        def __init__(self, d, n, K=1):
            self.ctx, self.n, self.K, self.graph = Ctx(d), n, K, None
            self.blocked, self.staged, self.sm_lanes, self.unstaged = 0, False, 0, False
            self._retired_graphs, self._keep = [], None

    e = Eng(50, 65536)
    e._regime(60.0)                                   # first steps of a run from the prior: redraws dominate
    assert e.staged and e.blocked == 0 and e.ctx.opts[OPT_STAGED_REDRAW] == 1 and e.ctx.opts.get(OPT_BLOCKED, 0) == 0
    e._regime(6.0)                                    # the walker's probe is the TRUE mean: it keeps the step down to 5
    assert e.staged and e.blocked == 0
    e._regime(4.0)
    assert not e.staged and e.blocked >= 1 and e.ctx.opts[OPT_STAGED_REDRAW] == 0
    assert e.blocked == 4 and e.ctx.opts[OPT_BLOCKED] == 4     # expected lists 49 152, 36 864, 27 648 >= 24 576 > 20 736
    e._regime(3.4)                                    # the blocked path's probe is the geometric estimate: stays up to 3.5
    assert not e.staged and e.blocked >= 1
    e._regime(3.6)
    assert e.staged and e.blocked == 0
    e._regime(1.0)
    assert not e.staged and e.blocked == 1            # (nearly) every first attempt in bounds: one round
    g = e.graph = object()
    e._regime(1.3)                                    # a captured graph keeps its kernel, rounds included ...
    assert e.graph is g and not e.staged and e.blocked == 1 and not e._retired_graphs
    e._regime(90.0)                                   # ... until the rule asks for ANOTHER kernel: retired, captured again later
    assert e.graph is None and e.staged and e.blocked == 0 and e._retired_graphs[0][0] is g
    e.graph = g
    e._regime(60.0)
    assert e.graph is g and len(e._retired_graphs) == 1
    # many particles: rounds while the expected list fills the chip, at most 24
    big = Eng(32, 262144)
    big._regime(2.27)
    assert big.blocked == 5 and big.ctx.opts[OPT_BLOCKED] == 5
    big.blocked = 1
    big._regime(3.4)
    assert 5 < big.blocked <= 24
    # n_dim >= 64: the crossovers sit higher (estimate 4.5, true mean 8)
    wide = Eng(100, 131072)
    wide._regime(7.0)
    assert wide.blocked >= 1 and not wide.staged
    wide._regime(4.4)
    assert wide.blocked >= 1
    wide._regime(4.6)
    assert wide.staged and wide.blocked == 0
    wide._regime(8.5)
    assert wide.staged
    wide._regime(7.9)
    assert wide.blocked >= 1 and not wide.staged
    # several modes: neither; the multi-lane kernel, un-staged while redraws dominate (hysteresis 8 / 4)
    k4 = Eng(32, 262144, K=4)
    k4._regime(20.0)
    assert k4.blocked == 0 and not k4.staged and k4.unstaged and k4.ctx.opts[OPT_ML_UNSTAGED] == 1
    k4._regime(5.0)
    assert k4.unstaged
    k4._regime(3.0)
    assert not k4.unstaged
    # d <= 16 has one kernel
    small = Eng(10, 1 << 20)
    small._regime(3.0)
    assert small.ctx.opts == {}
    monkeypatch.setenv("TEMPEST_AMD_SCREEN", "1")
    # ---- with the screened batches (default): their time is flat in the attempt count, so the crossovers sit lower at high
    # n_dim and higher at 32-D (tools/regime_sweep.py), and the blocked kernel runs more, shorter rounds (the straggler pass is
    # a screened launch over the list)
    s100 = Eng(100, 131072)
    s100._regime(60.0)
    assert s100.staged and s100.blocked == 0 and s100.ctx.opts[OPT_SCREEN] == 1
    s100._regime(5.1)
    assert s100.staged
    s100._regime(4.9)                                 # true mean below 5: blocked rounds
    assert not s100.staged and 1 <= s100.blocked <= 6
    s100._regime(2.9)                                 # now the geometric estimate: stays up to 3
    assert not s100.staged and s100.blocked == 6
    s100._regime(3.1)
    assert s100.staged and s100.blocked == 0
    s50 = Eng(50, 65536)                             # 33..63-D, with the list rounds fanned out: estimate 3.4 / true mean 5.5
    s50._regime(30.0)
    assert s50.staged and s50.blocked == 0
    s50._regime(5.6)
    assert s50.staged
    s50._regime(5.4)
    assert not s50.staged and 2 <= s50.blocked <= 8
    s50._regime(3.3)
    assert not s50.staged and s50.blocked >= 2
    s50._regime(3.5)
    assert s50.staged and s50.blocked == 0
    s50._regime(1.3)                                  # one try per round above 32-D; a fanned-out second round empties the list
    assert not s50.staged and 2 <= s50.blocked <= 4
    s32 = Eng(32, 262144)
    s32._regime(14.0)
    assert s32.staged
    s32._regime(12.0)                                 # 32-D: the screened batches only win above ~13 attempts per particle
    assert not s32.staged and s32.blocked == 12
    s32._regime(1.05)                                 # two attempts per round in place: lists 594, 1: two rounds
    assert s32.blocked == 2
    s32._regime(8.1)
    assert s32.staged
    # several modes: the matrix-core rounds over mode-pure tiles below ~8 estimated attempts, the screened batches mode by mode
    # above (round 5; the multi-lane kernel only with the screen off: previous test)
    m4 = Eng(32, 262144, K=4)
    m4._regime(20.0)
    assert m4.blocked == 0 and m4.staged
    m4._regime(12.0)
    assert m4.blocked == 12 and not m4.staged
    m4._regime(1.2)
    assert 2 <= m4.blocked <= 4 and not m4.staged
    m4._regime(8.5)
    assert m4.blocked == 0 and m4.staged
    m80 = Eng(32, 262144, K=80)                        # more modes than the mode tables hold: the multi-lane kernel
    m80._regime(20.0)
    assert m80.blocked == 0 and not m80.staged and m80.unstaged


def test_regime_rule_does_not_oscillate(monkeypatch):
    """a redraw probe that sits on a threshold of StepEngine._regime and crosses it at every reading (the blocked
    rounds report the geometric estimate, the screened batches the true mean: 3.1 >= 3.0 sends the step to the batches, whose 4.9 <
    5.0 sends it back) must not retire and re-capture the step's graph at every step: after two free switches the next ones wait
    4, 8, ... 64 readings.  Each threshold crossed ONCE, in either direction, switches at once; the debugging switches are read
    when the engine is built, not per step."""
    from tempest_amd import mcmc
    for var in ("TEMPEST_AMD_SCREEN", "TEMPEST_AMD_STAGED", "TEMPEST_AMD_BLK_MFMA", "TEMPEST_AMD_BLK_FAN", "TEMPEST_AMD_SM_LANES"):
        monkeypatch.delenv(var, raising=False)

    class Ctx:
        def __init__(self, d):
            self.n_dim, self.opts, self.calls = d, {}, 0

        def set_option(self, k, v):
            self.opts[k] = v
            self.calls += 1

    class Eng:
        _regime = mcmc.StepEngine._regime

        def __init__(self, d, n, K=1):
            self.ctx, self.n, self.K, self.graph = Ctx(d), n, K, None
            self.blocked, self.staged, self.sm_lanes, self.unstaged = 0, False, 0, False
            self._retired_graphs, self._keep = [], None
            self._opts = mcmc.RegimeOptions.from_env()
            self._since, self._dwell, self._quick = 1 << 30, 0, 0

    for d, n in ((100, 131072), (50, 65536), (32, 262144)):
        up, down, _, _ = mcmc.regime_band("screened", d)
        # one crossing each way switches immediately, in both directions, at exactly the tabulated numbers
        e = Eng(d, n)
        e._regime(60.0)
        assert e.staged and not e.blocked
        e._regime(down + 0.01)
        assert e.staged
        e._regime(down - 0.01)
        assert e.blocked >= 1 and not e.staged
        e._regime(up - 0.01)
        assert e.blocked >= 1
        e._regime(up + 0.01)
        assert e.staged and not e.blocked
        # the probe flips at every reading: count the kernel switches (= graph retirements) over 400 readings
        e = Eng(d, n)
        e._regime(60.0)
        switches, kinds = 0, []
        for _ in range(400):
            e.graph = object()                     # the step has been captured again since the last reading
            was = (e.blocked > 0, e.staged)
            e._regime(down - 0.01 if e.staged else up + 0.01)
            now = (e.blocked > 0, e.staged)
            if now != was:
                switches += 1
                assert e.graph is None and e._retired_graphs      # a switch retires the graph ...
            else:
                assert e.graph is not None                         # ... and nothing else does
            kinds.append(now)
        assert 4 <= switches <= 12, (d, switches)                  # 2 free + waits of 4, 8, 16, 32, 64, 64, ... readings
        assert len(e._retired_graphs) == switches
        # a run that has settled is free to switch again at once
        for _ in range(200):
            e._regime(1.05 if not e.staged else down - 0.01)
        assert e.blocked >= 1 and not e.staged
        e._regime(90.0)
        assert e.staged and not e.blocked
    # several modes: the same with the two-sided band of the matrix-core rounds
    sm = mcmc.REGIME_THRESHOLDS["several_modes"]
    e = Eng(32, 262144, K=4)
    e._regime(20.0)
    assert not e.blocked
    switches = 0
    for _ in range(400):
        was = e.blocked > 0
        e._regime(sm["up"] - 0.01 if not e.blocked else sm["down"] + 0.01)
        switches += (e.blocked > 0) != was
    assert switches <= 12
    # the environment is not consulted per step: a switch flipped AFTER construction changes nothing for this engine
    e = Eng(50, 65536)
    monkeypatch.setenv("TEMPEST_AMD_SCREEN", "0")
    e._regime(6.0)                                  # screened band of 50-D: 6.0 >= 5.5 keeps the batches; the walker band (5.0) too
    e._regime(5.2)                                  # screened: 5.2 < 5.5 -> blocked rounds; with the screen off it would have stayed
    assert e.blocked >= 1
    import inspect
    src = inspect.getsource(mcmc.StepEngine._regime)
    assert "os.environ" not in src and "import os" not in src


def test_virtual_shards_of_the_canonical_partition():
    """csrc/common.h: tph_vshards_for and its Python twin -- V depends on the particle count alone (never on the number of
    ranks): the largest of 48, 16, 12, 8, 6, 4, 3, 2, 1 with n % (256 V) == 0.  The BASELINE ensembles are powers of two (V = 16:
    1, 2, 4, 8 or 16 GPUs give the same bits); 3 * 2^k particles also admit 3, 6, 12, 24, 48."""
    from tempest_amd.device import vshards_for
    assert vshards_for(1048576) == 16 and vshards_for(2097152) == 16 and vshards_for(65536) == 16 and vshards_for(262144) == 16
    assert vshards_for(49152) == 48 and vshards_for(12288) == 48 and vshards_for(3 * 4096) == 48
    assert vshards_for(512) == 2 and vshards_for(1024) == 4 and vshards_for(256) == 1 and vshards_for(768) == 3
    assert vshards_for(1000) == 1 and vshards_for(4096) == 16 and vshards_for(2048) == 8
    for n in (256 * k for k in range(1, 200)):
        v = vshards_for(n)
        assert n % (256 * v) == 0 and v in (48, 16, 12, 8, 6, 4, 3, 2, 1)
        assert all(n % (256 * w) != 0 for w in (48, 16, 12, 8, 6, 4, 3, 2, 1) if w > v)
    # the C side exports the same rule through the bench digest only; the kernels are pinned by tests/test_distributed.py (GPU)
