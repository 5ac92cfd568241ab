"""Weighted Gaussian-mixture clustering of the trimmed history (reference: tempest/cluster.py).

`HierarchicalGaussianMixture` splits clusters 1 -> 2 while the BIC gain of a two-component weighted GMM over a
single Gaussian exceeds `threshold_modifier * n_params * log(ESS)` and both children keep `min_points` points
(cluster.py:420-520).  The passes over the data (E-step responsibilities and log-likelihood sums, M-step moment
sums, label prediction, k-means++ seeding distances) are HIP kernels (csrc/cluster.hip, csrc/modes.hip) on a
compact [0,1]-normalised SoA working set; the EM control flow, the BIC bookkeeping and the d x d inverses /
log-determinants are host logic, as in the reference.

Differences by design (documented in DESIGN.md):
  * the reference re-seeds NumPy's GLOBAL RNG with 42 inside every GaussianMixture.fit; here the seeding
    draws come from a private counter-based stream keyed the same way for every fit (equally deterministic, no side
    effect on the sampler's randomness);
  * unchanged clusters are not re-fitted on later split rounds (the reference re-fits them with the same seed and gets
    the same answer);
  * working sets larger than `max_points` are thinned by systematic resampling (uniform weights afterwards).
"""
import math
from typing import Optional

import numpy as np

from ._philox_host import uniform_scalar

TAG_CLUSTER = 9
LOG2PI = math.log(2.0 * math.pi)


def _inv_logdet(cov, reg):
    """(precision, logdet) of cov + reg I; falls back to reg I like cluster.py:186-193 on failure."""
    d = cov.shape[0]
    c = cov + np.eye(d) * reg
    try:
        L = np.linalg.cholesky(c)
    except np.linalg.LinAlgError:
        c = np.eye(d) * reg
        L = np.linalg.cholesky(c)
    return np.linalg.inv(c), 2.0 * float(np.sum(np.log(np.diag(L))))


def _pack_params(terms, means, covs, reg):
    """-> (K, 2 + d + d*d) array: [term, mean, precision, logdet] per component."""
    K, d = means.shape
    out = np.empty((K, 2 + d + d * d))
    for k in range(K):
        P, ld = _inv_logdet(covs[k], reg)
        out[k, 0] = terms[k]
        out[k, 1:1 + d] = means[k]
        out[k, 1 + d:1 + d + d * d] = P.reshape(-1)
        out[k, 1 + d + d * d] = ld
    return out


class _WorkingSet:
    """Compact SoA data (d, M) on the device + weights + per-point cluster labels."""

    def __init__(self, ctx, X, sw, n_components=2):
        import torch
        self.ctx, self.X, self.sw = ctx, X, sw
        self.M = X.shape[1]
        self.labels = torch.zeros(self.M, dtype=torch.int32, device=ctx.device)
        self.wr = ctx.empty(max(2, int(n_components)), self.M)
        self.stats = ctx.empty(3)
        self.tmp_labels = torch.empty(self.M, dtype=torch.int32, device=ctx.device)

    def to_dev(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.ctx.device)


class GaussianMixture:
    """Weighted EM for a Gaussian mixture on the device (cluster.py:5-340; covariance_type full | tied | diag | spherical: the
    E-step kernel takes full precision matrices, the M-step reduces the weighted scatter matrices to the stored form).  Host-array `fit`/`predict`/`bic`
    mirror the reference's interface; the hierarchical model drives `_fit_ws` on a shared working set."""

    device_paced_max_rows = 16384      # working sets up to this size run the EM loop device-paced (tph_gmm_em_*), see _fit_ws

    def __init__(self, n_components=1, covariance_type="full", max_iter=1000, n_init=1, tol=1e-3, reg_covar=1e-6,
                 random_state=None):
        if covariance_type not in ("full", "tied", "diag", "spherical"):
            raise ValueError(f"unknown covariance_type {covariance_type!r}")
        self.n_components = n_components
        self.covariance_type = covariance_type
        self.max_iter = max_iter
        self.n_init = n_init
        self.tol = tol
        self.reg_covar = reg_covar
        self.random_state = random_state
        self.weights_ = self.means_ = self.covariances_ = None
        self.converged_ = False
        self.n_iter_ = 0
        self.lower_bound_ = None
        self._ll_unweighted = None
        self._n_members = None

    # ------------------------------------------------------------------ device EM on a working set
    def _m_step(self, ws, K):
        """weights, means, covariances from wr = responsibilities x sample weights (cluster.py:200-235)."""
        ctx = ws.ctx
        d = ctx.n_dim
        import torch
        # the means are formed on the device (the same two IEEE operations as on the host) and feed the covariance kernels
        # directly: ONE device-to-host copy per M-step instead of a round trip per component and moment
        sums = torch.stack([ctx.x_weighted_sums(ws.X, ws.wr[k]) for k in range(K)])            # (K, 1 + d)
        means_dev = (sums[:, 1:] / (sums[:, :1] + 1e-10)).contiguous()
        covs_dev = torch.stack([ctx.x_weighted_cov(ws.X, ws.wr[k], means_dev[k]) for k in range(K)])   # (K, d * d)
        host = torch.cat([sums.reshape(-1), means_dev.reshape(-1), covs_dev.reshape(-1)]).cpu().numpy()
        tot = host[: K * (1 + d)].reshape(K, 1 + d)[:, 0].copy()
        means = host[K * (1 + d): K * (1 + d) + K * d].reshape(K, d).copy()
        scatter = host[K * (1 + d) + K * d:].reshape(K, d, d)          # sum_i wr_ki (x_i - mu_k)(x_i - mu_k)^T
        return tot / tot.sum(), means, self._covariances_from_scatter(scatter, tot, ws.M)

    def _covariances_from_scatter(self, scatter, tot, n_samples):
        """cluster.py:215-252: the stored form of each covariance type, from the weighted scatter matrices."""
        K, d = scatter.shape[0], scatter.shape[1]
        if self.covariance_type == "full":
            return scatter / (tot[:, None, None] + 1e-10)
        if self.covariance_type == "tied":                     # one (d, d) matrix; the reference divides by the row count
            return scatter.sum(axis=0) / n_samples
        diag = np.einsum("kjj->kj", scatter)
        if self.covariance_type == "diag":                     # (K, d)
            return diag / (tot[:, None] + 1e-10)
        return diag.sum(axis=1) / (tot * d + 1e-10)            # spherical: (K,)

    def _full(self, covs, d):
        """(K, d, d) matrices of the stored covariances (cluster.py:254-264)."""
        K = self.n_components
        if self.covariance_type == "full":
            return covs
        if self.covariance_type == "tied":
            return np.broadcast_to(covs, (K, d, d))
        if self.covariance_type == "diag":
            return np.stack([np.diag(c) for c in covs])
        return np.stack([np.eye(d) * c for c in covs])

    def _seed_row(self, ws, prob, u):
        """X[searchsorted(cumsum(prob), u * total)] (cluster.py:142-157)."""
        ctx = ws.ctx
        cdf = ctx.cdf(prob)
        r = u * float(cdf[-1].item())
        idx = ctx.resample_systematic(cdf, 1, r, renorm=1.0)       # #{cum < r} = np.searchsorted(cumsum, r)
        return ws.X[:, int(idx.item())].cpu().numpy()

    def _fit_ws(self, ws, swc, label, fit_id=0):
        """EM on the members of `label` with within-cluster normalised weights `swc` (zero elsewhere)."""
        ctx = ws.ctx
        d = ctx.n_dim
        K = self.n_components
        seed = 42 if self.random_state is None else int(self.random_state)
        best = None
        for init in range(self.n_init):
            # weighted k-means++ seeding (cluster.py:135-157)
            means = np.zeros((K, d))
            means[0] = self._seed_row(ws, swc, uniform_scalar(seed, fit_id, TAG_CLUSTER, init, 0))
            for k in range(1, K):
                p = _pack_params(np.zeros(k), means[:k], np.tile(np.eye(d), (k, 1, 1)), 0.0)
                ctx.gmm_estep(ws.X, swc, ws.labels, label, ws.to_dev(p), k, 1, wr=ws.wr)
                means[k] = self._seed_row(ws, ws.wr[0], uniform_scalar(seed, fit_id, TAG_CLUSTER, init, k))
            # initial responsibilities exp(-|x-mu_k|^2/2), normalised without epsilon (cluster.py:159-164)
            p = _pack_params(np.zeros(K), means, np.tile(np.eye(d), (K, 1, 1)), 0.0)
            p[:, -1] = -d * LOG2PI
            ctx.gmm_estep(ws.X, swc, ws.labels, label, ws.to_dev(p), K, 0, eps=0.0, wr=ws.wr, stats=ws.stats)
            if self.covariance_type == "full" and ws.M <= self.device_paced_max_rows:
                # Small working sets: device-paced loop (tph_gmm_em_*).  Parameters (incl. the d x d inverses), convergence test
                # and M-step stay on the device and the host reads 16 control words per BATCH of iterations: with the host in
                # every iteration an iteration cost 0.3 ms whatever the size -- 3.1 of the 3.7 s of a 1 000-particle run (10 500
                # iterations in 1 200 fits).  Large working sets keep the loop below: there an iteration is a millisecond of
                # kernels, the round trips are hidden behind them, and iterations enqueued past convergence would be the cost
                # (measured at 262 144 rows: 1.85 -> 2.1 s).
                state, off = ctx.gmm_em_state(K)
                ctx.gmm_em_begin(ws.X, K, ws.wr, state)
                # a single Gaussian is at its fixed point after the first M-step (the test passes at iteration 2: three passes);
                # two components take 9 ... 27 iterations (10th ... 90th percentile of the 1 000-particle run's 470 such fits)
                batch = 3 if K == 1 else 12
                while True:
                    ctx.gmm_em_run(ws.X, swc, ws.labels, label, K, ws.wr, state, self.reg_covar, self.tol, self.max_iter, batch)
                    host = state.cpu().numpy()            # the whole block (a few KB): one copy per batch, the last one is the result
                    ctl = host[:16]
                    if ctl[1] != 0.0:
                        break
                    batch = 3 if K == 1 else 6
                tail = host[off["weights"]:]
                weights = tail[:K].copy()
                means = tail[K:K + K * d].reshape(K, d).copy()
                covs = tail[K + K * d:].reshape(K, d, d).copy()
                lower, n_iter, stats = float(ctl[2]), int(ctl[3]), ctl[4:7].copy()
                if best is None or lower > best[0]:
                    best = (lower, weights, means, covs, n_iter, stats)
                continue
            weights, means, covs = self._m_step(ws, K)
            lower, n_iter, stats = -np.inf, 0, None
            for it in range(self.max_iter + 1):
                # E-step with the current parameters; its weighted log-likelihood is the lower bound the reference
                # computes right after the M-step that produced them (cluster.py:104-121)
                with np.errstate(divide="ignore"):
                    p = _pack_params(np.log(weights), means, self._full(covs, d), self.reg_covar)
                ctx.gmm_estep(ws.X, swc, ws.labels, label, ws.to_dev(p), K, 0, eps=1e-10, wr=ws.wr, stats=ws.stats)
                stats = ws.stats.cpu().numpy()
                if it > 0:
                    n_iter = it
                    if stats[0] - lower < self.tol or it == self.max_iter:
                        break
                    lower = stats[0]
                weights, means, covs = self._m_step(ws, K)
            if best is None or lower > best[0]:
                best = (lower, weights, means, covs, n_iter, stats)
        self.lower_bound_, self.weights_, self.means_, self.covariances_, self.n_iter_, stats = best
        self.converged_ = self.n_iter_ < self.max_iter
        self._ll_unweighted, self._n_members = float(stats[1]), int(stats[2])
        return self

    def _bic_ws(self):
        """cluster.py:330-340: -2 * (unweighted log-likelihood of the members) + n_parameters * log(n)."""
        d = self.means_.shape[1]
        K = self.n_components
        cov_par = {"full": K * d * (d + 1) / 2, "tied": d * (d + 1) / 2, "diag": K * d, "spherical": K}[self.covariance_type]
        n_par = (K - 1) + K * d + cov_par
        return -2.0 * self._ll_unweighted + n_par * math.log(self._n_members)

    def _predict_ws(self, ws, label, out):
        with np.errstate(divide="ignore"):
            p = _pack_params(np.log(self.weights_ + 1e-10), self.means_, self._full(self.covariances_, self.means_.shape[1]),
                             self.reg_covar)
        ws.ctx.gmm_estep(ws.X, None, ws.labels, label, ws.to_dev(p), self.n_components, 2, label_out=out)

    # ---------------------------------------------------------------------------- host-array API
    def _ws_from_host(self, X, sample_weight):
        import torch
        from .tools import _ctx
        X = np.asarray(X, dtype=np.float64)
        n, d = X.shape
        ctx = _ctx(d)
        sw = np.ones(n) if sample_weight is None else np.asarray(sample_weight, dtype=np.float64)
        if sw.shape[0] != n:
            raise ValueError("sample_weight must have the same length as X")
        sw = sw / sw.sum()
        Xt = torch.from_numpy(np.ascontiguousarray(X.T)).to(ctx.device)
        return _WorkingSet(ctx, Xt, torch.from_numpy(sw).to(ctx.device), self.n_components)

    def fit(self, X, sample_weight=None):
        ws = self._ws_from_host(X, sample_weight)
        return self._fit_ws(ws, ws.sw, 0)

    def predict(self, X):
        ws = self._ws_from_host(X, None)
        self._predict_ws(ws, 0, ws.tmp_labels)
        return ws.tmp_labels.cpu().numpy().astype(np.int64)

    def bic(self, X):
        ws = self._ws_from_host(X, None)
        with np.errstate(divide="ignore"):
            p = _pack_params(np.log(self.weights_), self.means_, self._full(self.covariances_, self.means_.shape[1]), self.reg_covar)
        ws.ctx.gmm_estep(ws.X, ws.sw, None, 0, ws.to_dev(p), self.n_components, 0, wr=ws.wr, stats=ws.stats)
        st = ws.stats.cpu().numpy()
        self._ll_unweighted, self._n_members = float(st[1]), int(st[2])
        return self._bic_ws()


class HierarchicalGaussianMixture:
    """BIC-driven recursive 1 -> 2 splitting (cluster.py:343-696) on the device."""

    def __init__(self, n_init=1, max_iterations=1000, min_points=None, threshold_modifier=1.0, covariance_type="full",
                 verbose=False, normalize=False, max_points: Optional[int] = 262144):
        if covariance_type != "full":
            raise NotImplementedError("only covariance_type='full' is on the GPU path")
        self.n_init = n_init
        self.max_iterations = max_iterations
        self.min_points = min_points
        self.covariance_type = covariance_type
        self.verbose = verbose
        self.normalize = normalize
        modifier = float(threshold_modifier)
        if modifier <= 0:
            raise ValueError("threshold_modifier must be positive.")
        self.threshold_modifier = modifier
        self.max_points = max_points
        self.labels_ = None
        self.cluster_centers_ = []
        self.cluster_covariances_ = []
        self.cluster_weights_ = []
        self.n_clusters_ = 0
        self._gmm_ready = False
        self._data_min = None
        self._data_max = None
        self._params_dev = None
        self._shift_dev = self._scale_dev = None
        self._remap = None

    # ------------------------------------------------------------------------------- fitting
    def _fit_ws(self, ws):
        """Split search on a normalised working set; fills labels and the per-cluster Gaussians."""
        import torch
        ctx = ws.ctx
        d = ctx.n_dim
        M = ws.M
        min_points = self.min_points if self.min_points is not None else 2 * d
        n_params = d + d * (d + 1) / 2 + 1
        clusters = [0]                    # label values, in the reference's list order
        next_label = 1
        cache = {}
        fit_id = 0
        sizes = {0: M}
        rounds = 0
        while rounds < self.max_iterations:
            rounds += 1
            best = None
            for lab in clusters:
                if sizes[lab] < min_points:
                    continue
                if lab not in cache:
                    mask = ws.labels == lab
                    swm = torch.where(mask, ws.sw, torch.zeros_like(ws.sw))
                    s = ctx.sum_sq_max(swm)
                    swc = swm / float(s[0])
                    n_eff = float(s[0] * s[0] / s[1])                          # cluster.py:381-392
                    threshold = self.threshold_modifier * n_params * math.log(n_eff)
                    parent = GaussianMixture(1, n_init=self.n_init, random_state=42)._fit_ws(ws, swc, lab, fit_id)
                    child = GaussianMixture(2, n_init=self.n_init, random_state=42)._fit_ws(ws, swc, lab, fit_id + 1)
                    fit_id += 2
                    improvement = parent._bic_ws() - child._bic_ws()
                    entry = {"improvement": improvement, "threshold": threshold, "child": child, "sizes": None}
                    if improvement > threshold:
                        child._predict_ws(ws, lab, ws.tmp_labels)
                        n1 = int((ws.tmp_labels == 1).sum().item())
                        entry["sizes"] = (sizes[lab] - n1, n1)
                        entry["split"] = ws.tmp_labels.clone()
                    cache[lab] = entry
                    if self.verbose:
                        print(f"Cluster {lab}: improvement={improvement:.2f}, threshold={threshold:.2f}")
                e = cache[lab]
                if e["improvement"] > e["threshold"] and e["sizes"] is not None and min(e["sizes"]) >= min_points:
                    if best is None or e["improvement"] > cache[best]["improvement"]:
                        best = lab
            if best is None:
                break
            e = cache.pop(best)
            a, b = next_label, next_label + 1
            next_label += 2
            ws.labels = torch.where(e["split"] == 0, torch.full_like(ws.labels, a),
                                    torch.where(e["split"] == 1, torch.full_like(ws.labels, b), ws.labels))
            sizes[a], sizes[b] = e["sizes"]
            clusters.remove(best)
            clusters.extend([a, b])
        # final per-cluster Gaussians (cluster.py:522-552) and dense labels in list order
        lut = torch.zeros(next_label, dtype=torch.int32, device=ctx.device)
        for i, lab in enumerate(clusters):
            lut[lab] = i
        ws.labels = lut[ws.labels.long()].contiguous()
        K = len(clusters)
        means, covs, wts = np.zeros((K, d)), np.zeros((K, d, d)), np.zeros(K)
        total = float(ctx.sum_sq_max(ws.sw)[0])
        for i in range(K):
            swm = torch.where(ws.labels == i, ws.sw, torch.zeros_like(ws.sw))
            s0 = float(ctx.sum_sq_max(swm)[0])
            wts[i] = s0 / total
            if sizes[clusters[i]] >= d:
                g = GaussianMixture(1, n_init=self.n_init, random_state=42)._fit_ws(ws, swm / s0, i, 10_000 + i)
                means[i], covs[i] = g.means_[0], g.covariances_[0]
            else:
                ones = torch.where(ws.labels == i, torch.ones_like(ws.sw), torch.zeros_like(ws.sw))
                sm = ctx.x_weighted_sums(ws.X, ones).cpu().numpy()
                means[i], covs[i] = sm[1:] / sm[0], np.eye(d)
        return K, means, covs, wts

    def _finish(self, ctx, K, means, covs, wts, lo, hi):
        import torch
        d = means.shape[1]
        self.n_clusters_ = K
        self._means_norm, self._covs_norm = means, covs
        self.cluster_weights_ = wts
        if self.normalize:
            scale = hi - lo
            self.cluster_centers_ = [m * scale + lo for m in means]
            self.cluster_covariances_ = [c * np.outer(scale, scale) for c in covs]
        else:
            self.cluster_centers_ = [m for m in means]
            self.cluster_covariances_ = [c for c in covs]
        self._gmm_ready = K > 0
        with np.errstate(divide="ignore"):
            p = _pack_params(np.log(wts + 1e-10), means, covs, 1e-6)                 # cluster.py:655-688
        self._params_dev = torch.from_numpy(p).to(ctx.device)
        if self.normalize:
            self._shift_dev = torch.from_numpy(np.ascontiguousarray(lo)).to(ctx.device)
            self._scale_dev = torch.from_numpy(np.ascontiguousarray(1.0 / (hi - lo + 1e-10))).to(ctx.device)
        else:
            self._shift_dev = self._scale_dev = None
        self._remap = None
        _ = d

    def _normalise_ws(self, ctx, X, sw):
        if not self.normalize:
            self._data_min = self._data_max = None
            return None, None
        _, rng = ctx.x_weighted_sums(X, sw, with_range=True)
        r = rng.cpu().numpy().reshape(-1, 2)
        lo, hi = r[:, 0].copy(), r[:, 1].copy()
        self._data_min, self._data_max = lo, hi
        import torch
        ctx.affine(X, torch.from_numpy(lo).to(ctx.device), torch.from_numpy(1.0 / (hi - lo + 1e-10)).to(ctx.device))
        return lo, hi

    def fit(self, X, sample_weight=None):
        """Host-array drop-in for cluster.py:420-570."""
        import torch
        from .tools import _ctx
        X = np.asarray(X, dtype=np.float64)
        n, d = X.shape
        sw = np.ones(n) if sample_weight is None else np.asarray(sample_weight, dtype=np.float64)
        if sw.shape[0] != n:
            raise ValueError("sample_weight must have the same length as X")
        ctx = _ctx(d)
        Xt = torch.from_numpy(np.ascontiguousarray(X.T)).to(ctx.device)
        swt = torch.from_numpy(sw).to(ctx.device)
        lo, hi = self._normalise_ws(ctx, Xt, swt)
        ws = _WorkingSet(ctx, Xt, swt)
        K, means, covs, wts = self._fit_ws(ws)
        self._finish(ctx, K, means, covs, wts, lo, hi)
        self.labels_ = ws.labels.cpu().numpy().astype(np.int64)
        return self

    def predict(self, X):
        import torch
        from .tools import _ctx
        X = np.asarray(X, dtype=np.float64)
        ctx = _ctx(X.shape[1])
        Xt = torch.from_numpy(np.ascontiguousarray(X.T)).to(ctx.device)
        return self.predict_device(Xt, ctx).cpu().numpy().astype(np.int64)

    def predict_proba(self, X):
        """(n, n_clusters_) membership probabilities (cluster.py:602-696): softmax over k of
        log(weight_k + 1e-10) + log N(x; mean_k, cov_k + 1e-6 I) in normalised coordinates, from the packed parameters the
        E-step kernel uses (any number of clusters; a few tensor operations on the device -- not a hot path)."""
        import torch
        from .tools import _ctx
        if not getattr(self, "_gmm_ready", False) or not self.cluster_centers_:
            raise ValueError("The model has not been fitted yet.")
        X = np.asarray(X, dtype=np.float64)
        n, d, K = X.shape[0], X.shape[1], self.n_clusters_
        ctx = _ctx(d)
        Xn = torch.from_numpy(np.ascontiguousarray(X.T)).to(ctx.device)
        if self._shift_dev is not None:
            Xn = (Xn - self._shift_dev[:, None]) * self._scale_dev[:, None]
        par = self._params_dev.to(ctx.device)
        logp = torch.empty(K, n, dtype=torch.float64, device=ctx.device)
        for k in range(K):
            diff = Xn - par[k, 1:1 + d, None]
            maha = (diff * (par[k, 1 + d:1 + d + d * d].reshape(d, d) @ diff)).sum(dim=0)
            logp[k] = par[k, 0] - 0.5 * (maha + par[k, 1 + d + d * d] + d * LOG2PI)
        P = torch.softmax(logp, dim=0).T.cpu().numpy()
        if self._remap is not None:                      # components merged into the labels predict() hands out
            rm = self._remap.cpu().numpy()
            Q = np.zeros((n, int(rm.max()) + 1))
            for k in range(K):
                Q[:, rm[k]] += P[:, k]
            P = Q
        return P

    # ----------------------------------------------------------------------------- device path
    def predict_device(self, x_soa, ctx=None):
        """argmax_k log(weight_k + 1e-10) + log N(x; mean_k, cov_k + 1e-6 I) in normalised coordinates
        (cluster.py:572-599,642-696) for a (d, n) SoA tensor of raw unit-cube points -> int32 labels."""
        import torch
        if ctx is None:
            from .tools import _ctx
            ctx = _ctx(x_soa.shape[0])
        out = torch.empty(x_soa.shape[1], dtype=torch.int32, device=x_soa.device)
        ctx.gmm_estep(x_soa.contiguous(), None, None, 0, self._params_dev, self.n_clusters_, 2, label_out=out,
                      shift=self._shift_dev, scale=self._scale_dev)
        if self._remap is not None:
            out = self._remap[out.long()].contiguous()
        return out

    def fit_predict_device(self, state, w, thr, refit, rng):
        """Trainer hook (train.py:97-116): (re)fit on the trimmed history when asked, then label EVERY history row
        with predict(); returns (labels int32 over the history, number of non-empty clusters)."""
        import torch
        from .device import KEY_U
        ctx = state.ctx
        n_h = w.numel()
        comm = state.comm
        sharded = comm is not None and comm.active
        if refit or not self._gmm_ready:
            th = thr.cpu().numpy()
            n_keep = int(th[2])
            if sharded:
                # the working set is the GLOBAL kept set in the reference's history order, rebuilt on every rank (every rank
                # then runs the identical split search on identical data: same kernels, same numbers, same decisions)
                from .sharding import gather_rows_in_order
                if self.max_points is not None and n_keep > self.max_points:
                    cdf = ctx.cdf_global(w, thr[0:1])
                    tick = rng.next()
                    u0 = uniform_scalar(rng.seed, tick, TAG_CLUSTER)
                    pick = ctx.resample_select_global(cdf, self.max_points, 1, rng.seed, tick, u0=u0, pscale=float(th[1]))
                    slots = torch.nonzero(pick >= 0).reshape(-1)
                    rows = pick[slots].contiguous()
                    Xl = ctx.gather_u_affine(rows)[0] if rows.numel() else ctx.empty(ctx.n_dim, 0)
                    X = gather_rows_in_order(comm, slots, Xl).contiguous()
                    sw = torch.full((self.max_points,), 1.0 / self.max_points, dtype=torch.float64, device=ctx.device)
                else:
                    keep = w >= thr[0]
                    m_loc = int(keep.sum().item())
                    pos_all = ctx.cdf_global(keep.to(torch.float64))           # global 1-based rank of every kept row (exact)
                    if m_loc:
                        idx = ctx.compact_indices(w, thr[0:1], m_loc)
                        Xl, swl = ctx.gather_u_affine(idx, w=w)
                        pos = (pos_all[idx] - 1.0).long()
                        cols = torch.cat([Xl, swl.reshape(1, -1)], dim=0)
                    else:
                        pos = torch.empty(0, dtype=torch.int64, device=ctx.device)
                        cols = ctx.empty(ctx.n_dim + 1, 0)
                    full = gather_rows_in_order(comm, pos, cols)
                    X, sw = full[:-1].contiguous(), full[-1].contiguous()
            elif self.max_points is not None and n_keep > self.max_points:
                cdf = ctx.cdf(w, thr[0:1])
                u0 = uniform_scalar(rng.seed, rng.next(), TAG_CLUSTER)
                idx = ctx.resample_systematic(cdf, self.max_points, u0, renorm=float(th[1]))
                X, _ = ctx.gather_u_affine(idx)
                sw = torch.full((self.max_points,), 1.0 / self.max_points, dtype=torch.float64, device=ctx.device)
            else:
                idx = ctx.compact_indices(w, thr[0:1], n_keep)
                X, sw = ctx.gather_u_affine(idx, w=w)
            lo, hi = self._normalise_ws(ctx, X, sw)
            ws = _WorkingSet(ctx, X, sw)
            K, means, covs, wts = self._fit_ws(ws)
            self._finish(ctx, K, means, covs, wts, lo, hi)
        ptr, ld = ctx.history_ptr(KEY_U)
        labels = torch.empty(n_h, dtype=torch.int32, device=ctx.device)
        ctx.gmm_estep(ptr, None, None, 0, self._params_dev, self.n_clusters_, 2, label_out=labels,
                      shift=self._shift_dev, scale=self._scale_dev, n=n_h, ld=ld)
        # modes are built for the labels that actually occur among the kept rows (np.unique, modes.py:183)
        kept = labels[w >= thr[0]]
        occ = torch.bincount(kept.long(), minlength=self.n_clusters_).to(torch.float64)
        if sharded:
            comm.all_reduce_sum(occ)
        present = occ > 0
        if bool(present.all()):
            self._remap = None
            return labels, self.n_clusters_
        remap = (torch.cumsum(present.to(torch.int32), 0) - 1).clamp(min=0).to(torch.int32)
        self._remap = remap
        return remap[labels.long()].contiguous(), int(present.sum().item())
