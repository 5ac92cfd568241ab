"""`fit_mvstud` with the reference's signature (tempest/student.py:6-116), evaluated on the device.

What the reference's EM returns with the NumPy / SciPy versions it pins: its first pass finds
`func0(1e300) >= 0`, sets nu = inf and returns the START values -- the per-dimension median, the covariance
`np.cov(data.T) (n - 1) / n + diag(var) / n`, ridged by `max(1e-6, 1e-6 |tr|)` only when its Cholesky factorisation fails
(student.py:60-64) -- so that is what this function computes by default (K11, `tph_fit_modes`: histogram-select median, centred
second moments, the same ridge rule).  `tolerance` and `max_iter` are accepted for the signature; like in the reference they
have no effect on that result.

`em=True` (keyword-only, an EXTENSION -- the reference's loop never iterates) runs the EM of student.py:66-116 with a working
degrees-of-freedom update: the digamma equation is bracketed on [1e-2, 1e6] instead of evaluated at 1e300, where it rounds to
>= 0 for any data.  The O(n) work is on the device: delta_i = (x_i - mu)^T Sigma^-1 (x_i - mu) (`tph_gmm_estep`, one
component), the data terms of the digamma equation for 16 trial nu per pass (`tph_student_sums`), the weights
(`tph_student_weights`) and the weighted moments (`tph_x_weighted_sums` / `tph_x_weighted_cov`); the host keeps the d x d
algebra and the scalar root search.  Checked against `oracle.ps.fit_mvstud_em` (tests/test_student_gpu.py)."""
import ctypes as C

import numpy as np

NU_LO, NU_HI = 1e-2, 1e6


def _ridge(S):
    try:
        np.linalg.cholesky(S)
        return S
    except np.linalg.LinAlgError:
        return S + np.eye(S.shape[0]) * max(1e-6, 1e-6 * abs(np.trace(S)))


def em_on_device(ctx, x_ptr, ld, n_rows, counts, labels, label, mu, Sigma, tolerance=1e-6, max_iter=100):
    """The EM of student.py:66-116 on rows x[:, :n_rows] (dimension-major device array at `x_ptr`, leading dimension `ld`) with
    int32 multiplicities `counts` and an optional label filter; (mu, Sigma) are the start values.  -> (mu, Sigma, nu, iterations)."""
    import torch
    from scipy import special
    from ._lib import check
    from .cluster import _pack_params
    d, dev = ctx.n_dim, ctx.device
    lib = ctx.lib
    lab_ptr = labels.data_ptr() if labels is not None else None
    sel = counts if labels is None else torch.where(labels == label, counts, torch.zeros_like(counts))
    n = float(sel.sum().item())
    if n <= 0:
        return mu, Sigma, np.inf, 0
    ones = torch.ones(n_rows, dtype=torch.float64, device=dev)
    delta = torch.empty(n_rows, dtype=torch.float64, device=dev)
    v = torch.empty(n_rows, dtype=torch.float64, device=dev)
    sums = torch.empty(1 + d, dtype=torch.float64, device=dev)
    cov = torch.empty(d * d, dtype=torch.float64, device=dev)
    out = (C.c_double * 32)()

    def data_terms(nus):
        """(sum c log w / n, sum c w / n) for each trial nu: one pass over delta per 16 of them."""
        res = []
        for a in range(0, len(nus), 16):
            chunk = np.ascontiguousarray(nus[a:a + 16], dtype=np.float64)
            check(lib.tph_student_sums(ctx._ctx, delta.data_ptr(), counts.data_ptr(), lab_ptr, int(label), n_rows,
                                       chunk.ctypes.data_as(C.c_void_p), len(chunk), out), "tph_student_sums")
            res += [(out[2 * b] / n, out[2 * b + 1] / n) for b in range(len(chunk))]
        return res

    def func0(nu, terms):
        return (-special.psi(nu / 2) + np.log(nu / 2) + terms[0] - terms[1] + 1 + special.psi((nu + d) / 2) - np.log((nu + d) / 2))
    nu, last_nu, it = 20.0, 0.0, 0
    mu = np.asarray(mu, dtype=np.float64).copy()
    while abs(last_nu - nu) > tolerance and it < max_iter:
        it += 1
        Sigma = _ridge(Sigma)
        # delta_i through the mixture E-step kernel with ONE component (mode 1: sw * |x - mu|^2_P, sw = 1)
        p = torch.from_numpy(_pack_params(np.zeros(1), mu[None], Sigma[None], 0.0)).to(dev)
        ctx.gmm_estep(int(x_ptr), ones, None, 0, p, 1, 1, wr=delta, n=n_rows, ld=ld)
        last_nu = nu
        lo, hi = NU_LO, NU_HI
        f_lo, f_hi = (func0(x, t) for x, t in zip((lo, hi), data_terms([lo, hi])))
        if f_hi >= 0:
            return mu, Sigma, np.inf, it
        if f_lo <= 0:
            nu = lo
        else:
            # 16 trial values per pass over delta, geometric grid inside the bracket, to 1e-13 relative
            for _ in range(40):
                grid = np.exp(np.linspace(np.log(lo), np.log(hi), 18))[1:-1]
                fs = [func0(x, t) for x, t in zip(grid, data_terms(list(grid)))]
                k = next((i for i, f in enumerate(fs) if f <= 0), len(grid))
                lo, hi = (grid[k - 1] if k > 0 else lo), (grid[k] if k < len(grid) else hi)
                if hi - lo <= 1e-13 * hi:
                    break
            nu = 0.5 * (lo + hi)
        check(lib.tph_student_weights(ctx._ctx, delta.data_ptr(), counts.data_ptr(), lab_ptr, int(label), n_rows, float(nu), v.data_ptr()),
              "tph_student_weights")
        mu_dev = torch.from_numpy(mu).to(dev)
        check(lib.tph_x_weighted_cov(ctx._ctx, int(x_ptr), ld, n_rows, v.data_ptr(), mu_dev.data_ptr(), cov.data_ptr()), "tph_x_weighted_cov")
        check(lib.tph_x_weighted_sums(ctx._ctx, int(x_ptr), ld, n_rows, v.data_ptr(), sums.data_ptr(), None), "tph_x_weighted_sums")
        s = sums.cpu().numpy()
        Sigma = cov.cpu().numpy().reshape(d, d) / n          # about the OLD mean (student.py:96-98)
        Sigma = 0.5 * (Sigma + Sigma.T)
        mu = s[1:] / s[0]
    return mu, _ridge(Sigma), nu, it


def fit_mvstud(data, tolerance=1e-6, max_iter=100, *, em=False):
    """(mu (dim,), Sigma (dim, dim), nu) for `data` of shape (n, dim); nu = inf unless `em=True` (see the module docstring)."""
    import torch
    from .device import KEY_U
    from .tools import _ctx
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 2 or data.shape[0] < 1:
        raise ValueError("data must have shape (n, dim)")
    n, dim = data.shape
    ctx = _ctx(dim)
    ctx.history_load(data, None, np.zeros(n), [0.0], [0.0], [n])
    counts = torch.ones(n, dtype=torch.int32, device=ctx.device)
    means, covs, _, _, _ = ctx.fit_modes(counts)
    mu, Sigma = means.cpu().numpy()[0], covs.cpu().numpy()[0]
    if not em:
        return mu, Sigma, np.inf
    ptr, ld = ctx.history_ptr(KEY_U)
    mu, Sigma, nu, _ = em_on_device(ctx, ptr, ld, n, counts, None, 0, mu, Sigma, tolerance, max_iter)
    return mu, Sigma, nu
