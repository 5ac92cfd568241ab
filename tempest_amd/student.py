"""`fit_mvstud` with the reference's signature (tempest/student.py:6-116), evaluated on the device.

What the reference's EM returns with the NumPy / SciPy versions it pins: its first pass finds
`func0(1e300) >= 0`, sets nu = inf and returns the START values -- the per-dimension median, the covariance
`np.cov(data.T) (n - 1) / n + diag(var) / n`, ridged by `max(1e-6, 1e-6 |tr|)` only when its Cholesky factorisation fails
(student.py:60-64) -- so that is what this function computes (K11, `tph_fit_modes`: histogram-select median, centred second
moments, the same ridge rule).  `tolerance` and `max_iter` are accepted for the signature; like in the reference they have no
effect on the result."""
import numpy as np


def fit_mvstud(data, tolerance=1e-6, max_iter=100):
    """(mu (dim,), Sigma (dim, dim), nu = inf) for `data` of shape (n, dim)."""
    import torch
    from .tools import _ctx
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 2 or data.shape[0] < 1:
        raise ValueError("data must have shape (n, dim)")
    n, dim = data.shape
    ctx = _ctx(dim)
    ctx.history_load(data, None, np.zeros(n), [0.0], [0.0], [n])
    counts = torch.ones(n, dtype=torch.int32, device=ctx.device)
    means, covs, _, _, _ = ctx.fit_modes(counts)
    return means.cpu().numpy()[0], covs.cpu().numpy()[0], np.inf
