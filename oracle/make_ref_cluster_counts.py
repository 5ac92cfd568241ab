"""Generate tests/golden/ref_cluster_counts.json: the number of proposal modes K the REFERENCE uses in every iteration
of whole runs with clustering=True (tempest/steps/train.py:97-104, tempest/cluster.py:420-600), next to the run's
evidence and step totals.

TEST INFRASTRUCTURE ONLY.  Usage (scratch cwd, reference read-only, never on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python3 /root/repo/oracle/make_ref_cluster_counts.py [--workers 6] [--seeds 8] [--configs a,b]

K is read where the reference itself reports it: the `ModeStatistics` that `Trainer.run` returns (its progress bar shows
the same number).  Only data is stored: per run the list of K by iteration, beta by iteration, logZ, iterations, steps.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_ref_ensembles import mixture32, prior20, rosenbrock  # noqa: E402


def separable32(x, sep=6.0, sig=0.3):
    """Four equal-weight N(mu_k, sig^2 I) modes at (+-sep, +-sep, 0, ...) in 32-D: the modes are 40 sigma apart in the
    two coordinates that tell them apart (config 3: 16 sigma), the twin the reference's BIC search does split."""
    d = x.shape[1]
    mus = np.zeros((4, d))
    for k, (a, b) in enumerate([(-sep, -sep), (-sep, sep), (sep, -sep), (sep, sep)]):
        mus[k, 0], mus[k, 1] = a, b
    q = ((x[:, None, :] - mus[None]) ** 2).sum(axis=2)
    from scipy.special import logsumexp
    return logsumexp(-0.5 * q / sig ** 2, axis=1) - np.log(4.0) - 0.5 * d * np.log(2 * np.pi * sig ** 2)


def bimodal8(x, sep=5.0, sig=0.5):
    """Two equal-weight N(+-sep e_0, sig^2 I) modes in 8-D (few parameters per component: the BIC threshold is low)."""
    d = x.shape[1]
    q0 = ((x - np.eye(d)[0] * sep) ** 2).sum(axis=1)
    q1 = ((x + np.eye(d)[0] * sep) ** 2).sum(axis=1)
    return np.logaddexp(-0.5 * q0 / sig ** 2, -0.5 * q1 / sig ** 2) - np.log(2.0) - 0.5 * d * np.log(2 * np.pi * sig ** 2)


CONFIGS = {
    # name: (loglike, n_dim, kwargs, n_total)
    "c1_rosenbrock_cluster": (rosenbrock, 10, dict(n_particles=1000), 4096),
    "c3twin_mix32_n1024_cluster": (mixture32, 32, dict(n_particles=1024, clustering=True), 4096),
    "sep32_n2048_cluster": (separable32, 32, dict(n_particles=2048, clustering=True), 8192),
    "bimodal8_n512_cluster": (bimodal8, 8, dict(n_particles=512, clustering=True), 2048),
}


def run_one(job):
    name, seed = job
    import tempest
    from tempest.steps import train as ref_train
    loglike, n_dim, kw, n_total = CONFIGS[name]
    ks = []
    orig = ref_train.Trainer.run

    def run(self, weights):
        ms = orig(self, weights)
        ks.append(int(ms.K))
        return ms

    ref_train.Trainer.run = run
    try:
        np.random.seed(seed)
        t0 = time.time()
        s = tempest.Sampler(prior20, loglike, n_dim, vectorize=True, **kw)
        s.run(n_total=n_total, progress=False)
        wall = time.time() - t0
    finally:
        ref_train.Trainer.run = orig
    beta = np.asarray(s.state.get_history("beta"))
    steps = np.asarray(s.state.get_history("steps"))
    x, w, _ = s.posterior()
    mean = np.average(x, weights=w, axis=0)
    return dict(config=name, seed=seed, logz=float(s.evidence()[0]), iters=int(len(beta)), K=ks,
                beta=[float(b) for b in beta], steps=[int(v) for v in steps], wall_s=wall,
                mean01=[float(mean[0]), float(mean[1])])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--out", default="/root/repo/tests/golden/ref_cluster_counts.json")
    a = ap.parse_args()
    import multiprocessing as mp
    import tempest
    jobs = [(c, k) for c in a.configs.split(",") for k in range(a.seed0, a.seed0 + a.seeds)]
    out = {"reference": "minaskar/tempest " + tempest.__version__, "numpy": np.__version__,
           "host": "build container, 8 vCPU Xeon 2.1 GHz, 1 thread per run", "runs": []}
    if os.path.exists(a.out):
        old = json.load(open(a.out))
        out["runs"] = [r for r in old.get("runs", []) if (r["config"], r["seed"]) not in set(jobs)]
    with mp.Pool(a.workers) as pool:
        for r in pool.imap_unordered(run_one, jobs):
            out["runs"].append(r)
            print(r["config"], r["seed"], round(r["logz"], 3), r["iters"], "K:", r["K"], round(r["wall_s"], 1), flush=True)
            out["runs"].sort(key=lambda q: (q["config"], q["seed"]))
            json.dump(out, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    sys.exit(main())
