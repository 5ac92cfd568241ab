"""Where does the FIRST run of a process lose time against the second?  `cold_warm.py c2|c3 [tpcn|rwm]`: the same seeded run of a
BASELINE configuration twice in one process (a fresh Sampler each), every phase of every iteration timed with a device
synchronisation behind it; prints per phase the seconds of run 1, of run 2, and the iterations where run 1 lost the most.
(The synchronisations cost overlap: the two totals are compared with each other, not with an un-instrumented run.)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(which, kernel, dev):
    import torch
    import tempest_amd as tp
    if which == "c3":
        d, n = 32, 262144
        mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
        for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
            mus[k, 0], mus[k, 1] = a, b
        const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

        def loglike(x):
            q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
            return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
        return lambda: tp.Sampler(lambda u: 20 * u - 10, loglike, d, vectorize=True, n_particles=n, clustering=True, random_state=0,
                                  sample=kernel, backend="torch", batch_prior=True), n
    d, n = 50, 65536
    A = np.random.RandomState(1).randn(d, d)
    S = A @ A.T / d + 0.5 * np.eye(d)
    P = torch.from_numpy(np.linalg.inv(S)).to(dev)
    const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
    return lambda: tp.Sampler(lambda u: 20 * u - 10, lambda x: -0.5 * ((x @ P) * x).sum(dim=1) + const, d, vectorize=True,
                              n_particles=n, clustering=False, random_state=0, sample=kernel, backend="torch", batch_prior=True), n


def main():
    import torch
    which, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "tpcn")
    dev = torch.device("cuda", 0)
    make, n = build(which, kernel, dev)
    runs = []
    for rep in range(2):
        torch.cuda.synchronize(dev)
        t_c0 = time.perf_counter()
        s = make()
        core = s._core
        rows = []
        phases = ("reweighter", "trainer", "resampler", "mutator")
        cur = {}
        for name in phases:
            obj = getattr(core, name)
            orig = obj.run

            def wrapped(*a, _orig=orig, _name=name, **k):
                t0 = time.perf_counter()
                r = _orig(*a, **k)
                torch.cuda.synchronize(dev)
                cur[_name] = time.perf_counter() - t0
                return r
            obj.run = wrapped
        commit = core.state.commit_current_to_history

        def commit_timed(*a, **k):
            t0 = time.perf_counter()
            r = commit(*a, **k)
            torch.cuda.synchronize(dev)
            cur["commit"] = time.perf_counter() - t0
            rows.append(dict(cur))
            cur.clear()
            return r
        core.state.commit_current_to_history = commit_timed
        t_construct = time.perf_counter() - t_c0
        t0 = time.perf_counter()
        s.run(n_total=4 * n, progress=False)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        runs.append({"construct_s": t_construct, "wall_s": wall, "rows": rows, "logz": float(s.evidence()[0]),
                     "startup": s.startup_breakdown})
        del s
    a, b = runs
    keys = ("reweighter", "trainer", "resampler", "mutator", "commit")
    tot = {k: [round(sum(r.get(k, 0.0) for r in run["rows"]), 4) for run in (a, b)] for k in keys}
    diffs = []
    for i, (ra, rb) in enumerate(zip(a["rows"], b["rows"])):
        for k in keys:
            dlt = ra.get(k, 0.0) - rb.get(k, 0.0)
            if dlt > 0.002:
                diffs.append((round(dlt, 4), i + 1, k, round(ra.get(k, 0.0), 4), round(rb.get(k, 0.0), 4)))
    diffs.sort(reverse=True)
    print(json.dumps({"config": which, "kernel": kernel, "first": {"construct_s": round(a["construct_s"], 4), "run_s": round(a["wall_s"], 4)},
                      "second": {"construct_s": round(b["construct_s"], 4), "run_s": round(b["wall_s"], 4)},
                      "same_logz": a["logz"] == b["logz"], "phase_seconds_first_second": tot, "iterations": len(a["rows"]),
                      "largest_losses_of_the_first_run": [{"lost_s": d[0], "iteration": d[1], "phase": d[2], "first_s": d[3], "second_s": d[4]} for d in diffs[:14]],
                      "startup_breakdown_first": a["startup"]}))


if __name__ == "__main__":
    main()
