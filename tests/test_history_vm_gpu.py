"""The history's memory (csrc/ctx.hip; tempest/state_manager.py:356-416 appends one iteration after the other without bound): u
and x of a large history live in a mapped address range that grows IN PLACE -- memory mapped behind what is there, nothing
reallocated or copied -- and outgrowing the reservation moves the mappings, not the data.  Checked here through the C ABI: every
row that went in comes back out whatever path the growth took (plain -> mapped migration, in-place growth, a wider
reservation), the row-major mirror follows, and kernels give the same bits on a mapped history as on a plain one."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _append(ctx, rs, n, d, t):
    u = rs.rand(d, n)
    x = 20 * u - 10
    logl = -0.5 * (x ** 2).sum(axis=0) + t
    dev = ctx.device
    ctx.history_append(torch.from_numpy(u).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(logl).to(dev), 0.01 * t, -1.0 * t)
    return u, x, logl


def _check(ctx, parts):
    from tempest_amd.device import KEY_LOGL, KEY_U, KEY_X
    u = np.concatenate([p[0] for p in parts], axis=1)
    x = np.concatenate([p[1] for p in parts], axis=1)
    logl = np.concatenate([p[2] for p in parts])
    assert ctx.size == logl.size
    np.testing.assert_array_equal(ctx.history_read(KEY_U, soa=True), u)
    np.testing.assert_array_equal(ctx.history_read(KEY_X, soa=True), x)
    np.testing.assert_array_equal(ctx.history_read(KEY_LOGL), logl)
    return u, x, logl


def _gather_check(ctx, u, x, logl, seed):
    n = logl.size
    idx = np.random.RandomState(seed).randint(0, n, size=min(n, 70000))
    d = ctx.n_dim
    uo, xo, lo = ctx.empty(d, idx.size), ctx.empty(d, idx.size), ctx.empty(idx.size)
    ctx.gather(torch.from_numpy(idx).to(ctx.device), uo, xo, lo)
    np.testing.assert_array_equal(uo.cpu().numpy(), u[:, idx])
    np.testing.assert_array_equal(xo.cpu().numpy(), x[:, idx])
    np.testing.assert_array_equal(lo.cpu().numpy(), logl[idx])


@pytest.mark.parametrize("d", [3, 10])
def test_mapped_history_grows_in_place_and_keeps_every_row(d):
    from tempest_amd.device import OPT_HISTORY_VM, HipContext
    ctx = HipContext(d, 0)
    ctx.set_option(OPT_HISTORY_VM, 2)                # mapped from the first row on
    rs = np.random.RandomState(d)
    parts = [_append(ctx, rs, 1000, d, 0)]
    m = ctx.history_memory()
    assert m["mapped"] == 1 and m["rows_backed"] >= 1000 and m["rows_reserved"] >= m["rows_backed"] and m["copies"] == 0
    reserved0 = m["rows_reserved"]
    _check(ctx, parts)
    for t in range(1, 6):                            # past the first reservation (4 x the first request, rounded to the granule)
        parts.append(_append(ctx, rs, 150000 + 1111 * t, d, t))
        u, x, logl = _check(ctx, parts)
        _gather_check(ctx, u, x, logl, t)            # through the row-major mirror, which grows the same way
    m = ctx.history_memory()
    assert m["mapped"] == 1 and m["copies"] == 0, m  # not one copy of the history
    assert m["growth_steps"] >= 2 and m["rows_backed"] >= ctx.size
    assert m["rows_reserved"] > reserved0 and m["rereservations"] >= 1, m      # the mappings moved to a wider range
    from tempest_amd import _lib
    assert m["mirror_rows"] >= ctx.size and m["mirror_drops"] == 0, (m, _lib.load().tph_last_error())
    ctx.close()


def test_plain_history_migrates_into_a_mapped_range():
    """A plain history moves into a mapped range with ONE copy (here on request; by itself when it is asked for >= 16 GB or when
    the next doubling would not fit) and grows without copies from there."""
    from tempest_amd.device import OPT_HISTORY_VM, HipContext
    d = 4
    ctx = HipContext(d, 0)
    rs = np.random.RandomState(1)
    parts = [_append(ctx, rs, 200000, d, 0), _append(ctx, rs, 200000, d, 1)]
    m = ctx.history_memory()
    assert m["mapped"] == 0 and m["rows_backed"] >= 400000                     # small: a plain allocation
    u, x, logl = _check(ctx, parts)
    _gather_check(ctx, u, x, logl, 0)
    ctx.set_option(OPT_HISTORY_VM, 2)
    parts.append(_append(ctx, rs, 200000, d, 2))                               # has to grow: moves into a mapped range
    m = ctx.history_memory()
    assert m["mapped"] == 1 and m["rows_backed"] >= 600000, m
    copies = m["copies"]
    u, x, logl = _check(ctx, parts)
    _gather_check(ctx, u, x, logl, 1)
    for t in range(3, 8):
        parts.append(_append(ctx, rs, 300000, d, t))
    m = ctx.history_memory()
    assert m["copies"] == copies, m                                            # from there on it grows without copies
    u, x, logl = _check(ctx, parts)
    _gather_check(ctx, u, x, logl, 2)
    ctx.close()


def test_kernels_give_the_same_bits_on_a_mapped_history():
    """reweight triples, proposal fit and volume variation on the same rows held plainly and in a mapped range."""
    from tempest_amd.device import OPT_HISTORY_VM, HipContext
    d, n, T = 6, 40000, 5
    outs = []
    for mode in (0, 2):
        ctx = HipContext(d, 0)
        ctx.set_option(OPT_HISTORY_VM, mode)
        rs = np.random.RandomState(7)
        for t in range(T):
            _append(ctx, rs, n, d, t)
        assert ctx.history_memory()["mapped"] == (1 if mode else 0)
        trip = ctx.reweight_eval([0.0, 0.03, 0.2])
        w = ctx.weights(0.03, trip[1][0], trip[1][1])
        counts = (torch.arange(ctx.size, device=ctx.device) % 3 == 0).to(torch.int32) * 2
        means, covs, chol, inv, winv = ctx.fit_modes(counts, None, 1, ctx.size)
        vv = ctx.volume_variation(w)
        outs.append((np.asarray(trip), means.cpu().numpy(), covs.cpu().numpy(), float(vv)))
        ctx.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
