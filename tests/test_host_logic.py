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
