"""HipCallbacks (tempest_amd/hipcallbacks.py): user prior/likelihood as HIP device functions compiled into the MCMC
step.  CPU: the plugin compiles for gfx950 and exports its C ABI.  GPU: its kernels against NumPy, its fused Metropolis
kernel against libtempest_hip's tph_accept on the same inputs (bit-exact), and whole runs fused vs. not fused."""
import ctypes
import shutil

import numpy as np
import pytest

torch = pytest.importorskip("torch")

SRC = '''
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
'''

needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not __import__("os").path.exists("/opt/rocm/bin/hipcc"),
                                 reason="hipcc not available")


def rosen_np(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2, axis=1)


@needs_hipcc
def test_plugin_builds_and_exports_abi():
    from tempest_amd.hipcallbacks import build_plugin
    path = build_plugin(SRC, 4)
    assert path.exists() and build_plugin(SRC, 4) == path              # cached by content hash
    assert build_plugin(SRC, 6) != path                                # n_dim is part of the key
    lib = ctypes.CDLL(str(path))
    for sym in ("tphu_last_error", "tphu_n_dim", "tphu_abi", "tphu_prior", "tphu_like", "tphu_accept", "tphu_step", "tphu_run"):
        assert hasattr(lib, sym), sym
    assert lib.tphu_n_dim() == 4 and lib.tphu_abi() == 3


@needs_hipcc
def test_bad_source_reports_compiler_output():
    from tempest_amd._lib import TempestHipError
    from tempest_amd.hipcallbacks import build_plugin
    with pytest.raises(TempestHipError, match="hipcc failed"):
        build_plugin("__device__ void prior_transform(const double* u, double* x) { x[0] = nonsense; }\n"
                     "__device__ double log_likelihood(const double* x) { return 0.0; }", 2)
    import tempest_amd as tp
    with pytest.raises(ValueError, match="log_likelihood"):
        tp.HipCallbacks("__device__ void prior_transform(const double* u, double* x) {}", 2)


@pytest.mark.gpu
@needs_hipcc
def test_callbacks_match_numpy():
    import tempest_amd as tp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d, n = 6, 3001
    cb = tp.HipCallbacks(SRC, d)
    rng = np.random.RandomState(0)
    u = rng.rand(n, d)
    x = cb.prior_transform(u)                                          # NumPy in -> NumPy out
    np.testing.assert_array_equal(x, 20.0 * u - 10.0)
    np.testing.assert_allclose(cb.log_likelihood(x), rosen_np(x), rtol=1e-14)
    # the sampler's convention: (n, d) strided view of a (d, n) device buffer, tensors out, no copies in between
    us = torch.from_numpy(np.ascontiguousarray(u.T)).cuda()
    xt = cb.prior_transform(us.T)
    assert xt.shape == (n, d) and xt.T.is_contiguous()
    np.testing.assert_array_equal(xt.cpu().numpy(), x)
    lt = cb.log_likelihood(xt)
    np.testing.assert_array_equal(lt.cpu().numpy(), cb.log_likelihood(x))
    assert cb.prior_transform(us.T[0]).shape == (d,) and cb.log_likelihood(xt[0]).dim() == 0


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("kernel,K", [("tpcn", 1), ("rwm", 1), ("tpcn", 3)])
def test_fused_accept_is_bit_identical_to_library_accept(kernel, K):
    """tphu_accept(u') == tph_accept(u', x' = plugin prior(u'), l' = plugin loglike(x')): same rows accepted, same
    u / x / logl afterwards, same per-cluster sums."""
    import tempest_amd as tp
    from tempest_amd.device import HipContext, KERNEL_ID
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d, n = 4, 5000
    cb = tp.HipCallbacks(SRC, d)
    ctx = HipContext(d, device=0)
    g = torch.Generator().manual_seed(3)
    u = torch.rand(d, n, generator=g, dtype=torch.float64).cuda()
    up = (u.cpu() + 0.01 * torch.randn(d, n, generator=g, dtype=torch.float64)).clamp(0, 1).cuda()
    x = cb.prior_transform(u.T).T.contiguous()
    logl = cb.log_likelihood(x.T)
    mu, mup = torch.rand(n, generator=g, dtype=torch.float64).cuda() * 8, torch.rand(n, generator=g, dtype=torch.float64).cuda() * 8
    assign = torch.randint(0, K, (n,), generator=g, dtype=torch.int32).cuda() if K > 1 else None
    dof = torch.full((K,), 5.0, dtype=torch.float64).cuda()
    beta, seed, tick = 0.31, 99, 17

    a = [t.clone() for t in (u, x, logl)]
    sums_a = ctx.zeros(1 + K)
    xp = cb.prior_transform(up.T).T.contiguous()
    lp = cb.log_likelihood(xp.T)
    mu_a, mu_b = mu.clone(), mu.clone()          # in/out: accepted rows take the proposal's Mahalanobis form
    ctx.accept(kernel, beta, a[0], a[1], a[2], up, xp, lp, mu_a, mup, assign, K, dof, seed, tick, 0, sums_a)
    b = [t.clone() for t in (u, x, logl)]
    sums_b = ctx.zeros(1 + K)
    part = ctx.empty(((n + 255) // 256) * (1 + K))
    cb.accept(KERNEL_ID[kernel], beta, b[0], b[1], b[2], up, mu_b, mup, assign, K, dof, seed, tick, 0, sums_b, partials=part)
    torch.cuda.synchronize()
    assert torch.equal(mu_a, mu_b)
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb)
    assert torch.equal(sums_a, sums_b)
    assert 0 < sums_a[0].item() < n


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("kernel,bcs", [("tpcn", None), ("rwm", None), ("tpcn", ([0], [2]))])
def test_whole_step_kernel_equals_two_kernel_step(kernel, bcs):
    """tphu_step (proposal + callbacks + Metropolis update in one kernel) == tph_propose followed by tphu_accept: whole runs
    bit-identical, with strict, periodic and reflective coordinates."""
    import tempest_amd as tp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = 4
    out = []
    for whole in (True, False):
        cb = tp.HipCallbacks(SRC, d, whole_step=whole)
        kw = dict(periodic=bcs[0], reflective=bcs[1]) if bcs else {}
        s = tp.Sampler(cb.prior_transform, cb.log_likelihood, d, n_particles=700, vectorize=True, clustering=False,
                       random_state=9, sample=kernel, graph=False, **kw)
        s.run(n_total=2800, progress=False)
        out.append((s.evidence()[0], np.asarray(s.state.get_history("steps")), s.posterior()[0],
                    np.asarray(s.state.get_history("acceptance"))))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    np.testing.assert_array_equal(out[0][3], out[1][3])


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("kernel,bcs,n,groups", [("tpcn", None, 700, 0), ("rwm", None, 700, 0), ("tpcn", ([0], [2]), 700, 0),
                                                 ("tpcn", None, 5000, 3), ("rwm", None, 2049, 2)])
def test_run_in_one_launch_equals_step_by_step(kernel, bcs, n, groups):
    """tphu_run (every step of a mutation run, adaptation and stopping rule included, in ONE cooperative launch) == the same steps
    launched one by one (tphu_step + tph_adapt): whole sampler runs bit-identical -- evidence, steps per iteration, acceptance,
    posterior -- also with several 256-particle tiles per workgroup (`run_groups`) and with a ragged last tile."""
    import tempest_amd as tp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = 4
    out = []
    for persistent in (True, False):
        cb = tp.HipCallbacks(SRC, d, persistent=persistent)
        cb.run_groups = groups
        kw = dict(periodic=bcs[0], reflective=bcs[1]) if bcs else {}
        s = tp.Sampler(cb.prior_transform, cb.log_likelihood, d, n_particles=n, vectorize=True, clustering=False,
                       random_state=9, sample=kernel, graph=False, **kw)
        s.run(n_total=3 * n, progress=False)
        assert cb.persistent == persistent                      # (the device did not refuse the cooperative launch)
        out.append((s.evidence()[0], np.asarray(s.state.get_history("steps")), s.posterior()[0],
                    np.asarray(s.state.get_history("acceptance")), np.asarray(s.state.get_history("efficiency"))))
    assert out[0][0] == out[1][0]
    for k in (1, 2, 3, 4):
        np.testing.assert_array_equal(out[0][k], out[1][k])


@pytest.mark.gpu
@needs_hipcc
def test_run_in_one_launch_sums_partials_like_the_wide_adapt_kernel():
    """Above 1024 tiles tph_adapt sums the tile partials with 1024 threads; the one-launch run reproduces that order with its 256
    (four strided chains per thread, sixteen wave sums): 300 000 particles, same steps and acceptance bit for bit."""
    import tempest_amd as tp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d, n = 4, 300_000
    out = []
    for persistent in (True, False):
        cb = tp.HipCallbacks(SRC, d, persistent=persistent)
        s = tp.Sampler(cb.prior_transform, cb.log_likelihood, d, n_particles=n, vectorize=True, clustering=False,
                       random_state=3, graph=False)
        s.run(n_total=n, progress=False)
        out.append((s.evidence()[0], np.asarray(s.state.get_history("steps")), np.asarray(s.state.get_history("acceptance"))))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("graph", [False, True])
def test_fused_run_equals_unfused_run(graph):
    """A whole run with the fused step == the same plugin used as plain callbacks (library tph_accept)."""
    import tempest_amd as tp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = 4
    out = []
    for fused in (True, False):
        cb = tp.HipCallbacks(SRC, d, fused=fused)
        s = tp.Sampler(cb.prior_transform, cb.log_likelihood, d, n_particles=512, vectorize=True, clustering=False,
                       random_state=4, graph=graph)
        s.run(n_total=2048, progress=False)
        assert (s._core.callbacks.hip_plugin is cb) == fused
        out.append((s.evidence()[0], np.asarray(s.state.get_history("steps")), s.posterior()[0]))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    # and the answer is the Rosenbrock evidence (4-D: two independent 2-D factors of the README target)
    s = tp.Sampler(cb.prior_transform, cb.log_likelihood, d, n_particles=2048, vectorize=True, clustering=False,
                   random_state=1)
    s.run(n_total=8192, progress=False)
    # per 2-D factor: integral of exp(-10 (x^2-y)^2 - (x-1)^2) = pi / sqrt(10); prior volume 400
    truth = 2 * (np.log(np.pi / np.sqrt(10.0)) - np.log(400.0))
    assert abs(s.evidence()[0] - truth) < 0.25, (s.evidence()[0], truth)
