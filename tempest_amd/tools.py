"""Host-facing helpers with the reference's names and semantics (tempest/tools.py), evaluated by the
HIP kernels.  NumPy in, NumPy out; the steps themselves never go through these wrappers (they keep
everything on the device), these exist so code written against `tempest.tools` keeps working."""
import math
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

SQRTEPS = math.sqrt(float(np.finfo(np.float64).eps))

_CTX = {}


def _ctx(n_dim=1):
    """Scratch device context per dimension (lazily created, reused)."""
    from .device import HipContext
    c = _CTX.get(n_dim)
    if c is None:
        c = _CTX[n_dim] = HipContext(n_dim)
    c.use_current_stream()
    return c


def _to_dev(a, ctx, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64 if dtype is None else dtype))
    return t.to(ctx.device)


def effective_sample_size(weights: np.ndarray) -> float:
    """1 / sum (w / sum w)^2  (tools.py:120-135)."""
    w = np.asarray(weights, dtype=np.float64)
    if w.size == 0:
        return float("nan")
    c = _ctx()
    s = c.sum_sq_max(_to_dev(w, c))
    return float(s[0] * s[0] / s[1])


def compute_ess(logw: np.ndarray) -> float:
    """ESS fraction of log-weights (tools.py:138-156)."""
    lw = np.asarray(logw, dtype=np.float64)
    c = _ctx()
    c.history_load(None, None, lw, [0.0], [0.0], [lw.size])   # C_s = log n, v = logw - log n
    m, s1, s2 = c.reweight_eval([1.0])[0]
    return float(s1 * s1 / s2 / lw.size)


def increment_logz(logw: np.ndarray) -> float:
    """log sum exp(logw)  (tools.py:159-175)."""
    lw = np.asarray(logw, dtype=np.float64)
    c = _ctx()
    c.history_load(None, None, lw, [0.0], [0.0], [lw.size])   # C_s = log n, v = logw - log n
    m, s1, _ = c.reweight_eval([1.0])[0]
    return float(m + math.log(s1) + math.log(lw.size))


def trim_weights(samples: np.ndarray, weights: np.ndarray, ess: float = 0.99, bins: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    """Keep the smallest high-weight set whose ESS is >= ess of the total (tools.py:10-55).
    Like the reference this normalises `weights` in place."""
    weights /= np.sum(weights)
    c = _ctx()
    _, out = c.trim_threshold(_to_dev(weights, c), ess, bins, sync=True)
    mask = weights >= out[0]
    return samples[mask], weights[mask] / out[1]


def volume_variation(x, w=None) -> float:
    """CV of sqrt(det Cov) by the influence-function formula (tools.py:58-117)."""
    x = np.asarray(x, dtype=np.float64)
    n, d = x.shape
    if n < d + 1:
        return 1e10
    if w is None:
        w = np.ones(n)
    w = np.asarray(w, dtype=np.float64)
    w = w / np.sum(w)
    c = _ctx(d)
    c.history_load(x, None, np.zeros(n), [0.0], [0.0], [n])
    return device_volume_variation(c, _to_dev(w, c), n)


def device_volume_variation(ctx, w_dev, n_global, comm=None) -> float:
    """tools.py:58-117 on the device history of `ctx` with (normalised) device weights: one library call, the d x d rank
    rule / Cholesky / inverse included (tph_volume_variation), one host wait for the scalar.  With a communicator attached
    to the ctx the moments and the sum are all-reduced inside the library (`comm` is kept for the call sites' signature)."""
    d = ctx.n_dim
    if n_global < d + 1:
        return 1e10
    centre = None
    if d <= 12:
        # one pass for mean and covariance: moments about the previous call's mean; the library leaves the new mean in the
        # same buffer.  First call: any point inside the data will do -- the centre of the unit cube, on one GPU and on every
        # rank of a sharded run alike (the centre is part of the arithmetic: the same one keeps the statistic bitwise the same
        # on any number of ranks)
        centre = getattr(ctx, "_vv_centre", None)
        if centre is None:
            import torch
            centre = torch.full((d,), 0.5, dtype=torch.float64, device=ctx.device)
            ctx._vv_centre = centre
    return float(ctx.volume_variation(w_dev, centre))


def systematic_resample(size: int, weights: np.ndarray, random_state: Optional[int] = None) -> np.ndarray:
    """Systematic resampling (tools.py:178-228); the single uniform comes from NumPy's global stream
    exactly where the reference draws it."""
    if random_state is not None:
        np.random.seed(random_state)
    w = np.asarray(weights, dtype=np.float64)
    tot = float(np.sum(w))
    renorm = tot if abs(tot - 1.0) > SQRTEPS else 1.0
    u0 = np.random.random()
    c = _ctx()
    cdf = c.cdf(_to_dev(w, c))
    return c.resample_systematic(cdf, int(size), u0, renorm=renorm).cpu().numpy()


class ProgressBar:
    """tqdm-backed progress display (tools.py:231-267)."""

    def __init__(self, show: bool = True, initial: int = 0):
        from tqdm import tqdm
        self.progress_bar = tqdm(desc="Iter", disable=not show, initial=initial)
        self.info: Dict[str, Any] = dict()

    def update_stats(self, info: Dict[str, Any]) -> None:
        self.info = {**self.info, **info}
        self.progress_bar.set_postfix(ordered_dict=self.info)

    def update_iter(self) -> None:
        self.progress_bar.update(1)

    def close(self) -> None:
        self.progress_bar.close()


class FunctionWrapper(object):
    """Bind extra args/kwargs to the likelihood (tools.py:270-309)."""

    def __init__(self, f: Callable, args: Optional[List[Any]], kwargs: Optional[Dict[str, Any]]):
        self.f = f
        self.args = [] if args is None else args
        self.kwargs = {} if kwargs is None else kwargs

    def __call__(self, x) -> Any:
        return self.f(x, *self.args, **self.kwargs)
