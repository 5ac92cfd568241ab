"""Generate tests/golden/ref_ensembles.json by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Usage (scratch cwd, reference read-only, never on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python3 /root/repo/oracle/make_ref_ensembles.py [--workers 6] [--seeds 16]

Each entry is one `tempest.Sampler(...).run()` of the reference with `np.random.seed(k)` set
explicitly (the reference's `random_state` does not seed a fresh run, core.py:313-315).  Only
summary statistics are stored (logZ, iteration count, step totals, weighted moments): they define
the empirical sigma_ref of BASELINE.md section 2, against which the GPU build's logZ is gated.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402


def rosenbrock(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)


E2E_MEAN = np.array([2.0, -1.5, 0.5, 3.2, -2.8, 1.1, -0.7, 2.5, -1.2, 0.9])
E2E_VAR = np.array([1.0, 0.8, 1.2, 0.9, 1.1, 0.7, 1.3, 0.85, 1.15, 0.95])


def gauss_e2e(x):
    return -0.5 * np.sum((x - E2E_MEAN) ** 2 / E2E_VAR, axis=1) \
        - 0.5 * np.sum(np.log(2 * np.pi * E2E_VAR))


def c2_cov(d=50):
    A = np.random.RandomState(1).randn(d, d)
    return A @ A.T / d + 0.5 * np.eye(d)


_C2 = {}


def gauss_c2(x):
    if _C2.get("d") != x.shape[1]:
        _C2["d"] = x.shape[1]
        S = c2_cov(x.shape[1])
        _C2["P"] = np.linalg.inv(S)
        _C2["ld"] = np.linalg.slogdet(S)[1]
    d = x.shape[1]
    return -0.5 * np.einsum("ij,jk,ik->i", x, _C2["P"], x) - 0.5 * _C2["ld"] - 0.5 * d * np.log(2 * np.pi)


def prior20(u):
    return 20 * u - 10


def funnel(x):
    d = x.shape[1]
    v = x[:, 0]
    lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
    lr = np.sum(-0.5 * x[:, 1:] ** 2 * np.exp(-v)[:, None], axis=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
    return lv + lr


def prior_funnel(u):
    x = 600.0 * u - 300.0
    x[..., 0] = 30.0 * u[..., 0] - 15.0
    return x


def mixture32(x):
    """config 3: four equal-weight N(mu_k, 0.25 I) modes at (+-4, +-4, 0, ...) in 32-D."""
    d = x.shape[1]
    mus = np.zeros((4, d))
    for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
        mus[k, 0], mus[k, 1] = a, b
    q = ((x[:, None, :] - mus[None]) ** 2).sum(axis=2)
    from scipy.special import logsumexp
    return logsumexp(-0.5 * q / 0.25, axis=1) - np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25)


PRIORS = {"c5twin_funnel100_n4096": prior_funnel}

CONFIGS = {
    # name: (loglike, n_dim, kwargs, n_total)
    "c1_rosenbrock_cluster": (rosenbrock, 10, dict(n_particles=1000), 4096),
    "c1_rosenbrock_nocluster": (rosenbrock, 10, dict(n_particles=1000, clustering=False), 4096),
    "e2e_gauss10_n128": (gauss_e2e, 10, dict(n_particles=128, clustering=False, n_steps=1), 2048),
    "e2e_gauss10_n128_rwm_syst": (gauss_e2e, 10, dict(n_particles=128, clustering=False, sample="rwm",
                                                      resample="syst"), 2048),
    "c2twin_gauss50_n512_rwm": (gauss_c2, 50, dict(n_particles=512, clustering=False, sample="rwm"), 2048),
    # cheaper high-dimensional twins (the 50-D RWM twin spends hours in the reference's per-walker redraw loop)
    "gauss20_n256_tpcn": (gauss_c2, 20, dict(n_particles=256, clustering=False), 1024),
    "gauss20_n256_rwm": (gauss_c2, 20, dict(n_particles=256, clustering=False, sample="rwm"), 1024),
    # dynamic (volume-variation) beta schedule and boundary conditions, end to end
    "e2e_gauss10_n128_dynamic": (gauss_e2e, 10, dict(n_particles=128, clustering=False, volume_variation=0.5), 2048),
    "e2e_gauss10_n128_bc": (gauss_e2e, 10, dict(n_particles=128, clustering=False, periodic=[0, 3], reflective=[1, 4]), 2048),
    # config 2 with the tpCN kernel (its RWM variant spends hours in the reference's redraw loop) and config 3, small N
    "c2twin_gauss50_n512_tpcn": (gauss_c2, 50, dict(n_particles=512, clustering=False), 2048),
    "c3twin_mix32_n1024_cluster": (mixture32, 32, dict(n_particles=1024, clustering=True), 4096),
    # the parity twin of config 5 (100-D funnel) at N = 4096, reported as-is
    "c5twin_funnel100_n4096": (funnel, 100, dict(n_particles=4096, clustering=False), 4 * 4096),
}


def run_one(job):
    name, seed = job
    import tempest
    loglike, n_dim, kw, n_total = CONFIGS[name]
    np.random.seed(seed)
    t0 = time.time()
    s = tempest.Sampler(PRIORS.get(name, prior20), loglike, n_dim, vectorize=True, **kw)
    s.run(n_total=n_total, progress=False)
    wall = time.time() - t0
    x, w, _ = s.posterior()
    mean = np.average(x, weights=w, axis=0)
    var = np.average((x - mean) ** 2, weights=w, axis=0)
    beta = np.asarray(s.state.get_history("beta"))
    steps = np.asarray(s.state.get_history("steps"))
    n = kw["n_particles"]
    pms = int(np.sum(steps[beta > 0]) * n)
    return dict(config=name, seed=seed, logz=float(s.evidence()[0]), iters=int(len(beta)),
                calls=int(s.state.get_current("calls")), pms=pms, wall_s=wall,
                n_warm=int(np.sum(beta == 0)), acceptance_last=float(s.state.get_current("acceptance")),
                mean=[float(v) for v in mean], var=[float(v) for v in var])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--seeds", type=int, default=16)
    ap.add_argument("--seed0", type=int, default=0, help="first seed; earlier results of the same configs with other seeds are kept")
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--out", default="/root/repo/tests/golden/ref_ensembles.json")
    a = ap.parse_args()
    import multiprocessing as mp
    import tempest
    jobs = [(c, k) for c in a.configs.split(",") for k in range(a.seed0, a.seed0 + a.seeds)]
    out = {"reference": "minaskar/tempest " + tempest.__version__, "numpy": np.__version__,
           "host": "build container, 8 vCPU Xeon 2.1 GHz, 1 thread per run", "runs": []}
    if os.path.exists(a.out):
        old = json.load(open(a.out))
        keep = [r for r in old.get("runs", []) if (r["config"], r["seed"]) not in set(jobs)]
        out["runs"] = keep
    with mp.Pool(a.workers) as pool:
        for r in pool.imap_unordered(run_one, jobs):
            out["runs"].append(r)
            print(r["config"], r["seed"], round(r["logz"], 3), r["iters"], round(r["wall_s"], 1), flush=True)
            json.dump(out, open(a.out, "w"), indent=0)
    summ = {}
    for c in CONFIGS:
        lz = np.array([r["logz"] for r in out["runs"] if r["config"] == c])
        if lz.size:
            summ[c] = dict(n=int(lz.size), logz_mean=float(lz.mean()), logz_std=float(lz.std(ddof=1)))
    out["summary"] = summ
    json.dump(out, open(a.out, "w"), indent=0)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    sys.exit(main())
