"""One BASELINE.json configuration end to end on one GPU (used under rocprofv3 for profiles/): `run_config.py c2|c3|c5 rwm|tpcn`
(c5: a 131 072-particle shard of config 5's 100-D funnel; sep / sep1: the separable 32-D four-mode twin of config 3 that the
reference's BIC search does split (oracle/make_ref_cluster_counts.py), 262 144 particles, with clustering -- K grows to 4 --
and without -- K = 1 --; TEMPEST_AMD_RUN_PARTICLES overrides the particle count)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main():
    import torch
    import tempest_amd as tp
    which, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "tpcn")
    dev = torch.device("cuda", 0)
    assert which in ("c2", "c3", "c5", "sep", "sep1")
    ks, phase = [], {}
    if which in ("sep", "sep1"):
        d, n = 32, int(__import__("os").environ.get("TEMPEST_AMD_RUN_PARTICLES", "262144"))
        mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
        for k, (a, b) in enumerate([(-6, -6), (-6, 6), (6, -6), (6, 6)]):
            mus[k, 0], mus[k, 1] = a, b
        const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.09))

        def loglike(x):
            q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
            return torch.logsumexp(-0.5 * q / 0.09, dim=1) + const
        def make():
            s = tp.Sampler(lambda u: 20 * u - 10, loglike, d, vectorize=True, n_particles=n, clustering=which == "sep", random_state=0,
                           sample=kernel, backend="torch", batch_prior=True,
                           split_threshold=float(__import__("os").environ.get("TEMPEST_AMD_RUN_SPLIT", "1.0")))
            # TEMPEST_AMD_RUN_MAX_POINTS: the clustering working set is thinned to that many rows (the documented `max_points` of
            # the device clustering, default 262 144): at a few thousand rows the split search does find the four modes, which is
            # how a K = 4 run at 262 144 particles is produced for the timing of the several-modes proposal path
            mp = __import__("os").environ.get("TEMPEST_AMD_RUN_MAX_POINTS")
            if mp and s._core.trainer.clusterer is not None:
                s._core.trainer.clusterer.max_points = int(mp)
            return s
        from tempest_amd.steps import mutate as mu_, train as tr
        orig_t, orig_m = tr.Trainer.run, mu_.Mutator.run

        def trun(self, w):
            t = time.perf_counter()
            ms = orig_t(self, w)
            torch.cuda.synchronize()
            phase["train"] = phase.get("train", 0.0) + time.perf_counter() - t
            ks.append(int(ms.K))
            return ms

        def mrun(self, ms):
            t = time.perf_counter()
            r = orig_m(self, ms)
            torch.cuda.synchronize()
            phase["mutate"] = phase.get("mutate", 0.0) + time.perf_counter() - t
            return r
        tr.Trainer.run, mu_.Mutator.run = trun, mrun
    elif which == "c3":       # 32-D four-mode Gaussian mixture, 262 144 particles, clustering (BASELINE config 3)
        d, n = 32, 262144
        mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
        for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
            mus[k, 0], mus[k, 1] = a, b
        const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

        def loglike(x):
            q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
            return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
        make = lambda: tp.Sampler(lambda u: 20 * u - 10, loglike, d, vectorize=True, n_particles=n, clustering=True, random_state=0,   # noqa: E731
                                  sample=kernel, backend="torch", batch_prior=True)
    elif which == "c5":     # 100-D Neal funnel, a 131 072-particle shard of BASELINE config 5's 2 097 152
        d, n = 100, int(__import__("os").environ.get("TEMPEST_AMD_RUN_PARTICLES", "131072"))
        scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
        shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

        def loglike(x):
            v = x[:, 0]
            lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
            lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
            return lv + lr
        make = lambda: tp.Sampler(lambda u: u * scale + shift, loglike, d, vectorize=True, n_particles=n, clustering=False,   # noqa: E731
                                  random_state=0, sample=kernel, backend="torch", batch_prior=True)
    else:
        d, n = 50, 65536
        A = np.random.RandomState(1).randn(d, d)
        S = A @ A.T / d + 0.5 * np.eye(d)
        P = torch.from_numpy(np.linalg.inv(S)).to(dev)
        const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
        make = lambda: tp.Sampler(lambda u: 20 * u - 10, lambda x: -0.5 * ((x @ P) * x).sum(dim=1) + const, d, vectorize=True,   # noqa: E731
                                  n_particles=n, clustering=False, random_state=0, sample=kernel, backend="torch", batch_prior=True)
    # TEMPEST_AMD_RUN_REPEAT=2: the same run a second time in the same process (a new Sampler, same seed): the first pays the
    # process's one-off costs -- code objects, rocBLAS, allocator pools, graph captures --, the second is what a long job sees
    for rep in range(int(__import__("os").environ.get("TEMPEST_AMD_RUN_REPEAT", "1"))):
        ks.clear(); phase.clear()
        s = make()
        # TEMPEST_AMD_RUN_PHASES=1: the LAST run with a device synchronisation behind every phase, their seconds in the line
        phases = __import__("os").environ.get("TEMPEST_AMD_RUN_PHASES") == "1" and rep + 1 == int(__import__("os").environ.get("TEMPEST_AMD_RUN_REPEAT", "1"))
        s._core.profile = phases
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(n_total=4 * n, progress=False)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        steps = np.asarray(s.state.get_history("steps")); beta = np.asarray(s.state.get_history("beta"))
        print(f'{{"config": "{which}", "kernel": "{kernel}", "run_in_process": {rep + 1}, "n_dim": {d}, "n_particles": {n}, "logz": {s.evidence()[0]}, '
              f'"analytic_logz": {-d * np.log(20.0) if which != "c5" else -np.log(30.0) - 99 * np.log(600.0)}, "iterations": {len(beta)}, "mcmc_steps": {int(steps[beta > 0].sum())}, '
              f'"wall_s": {wall}, "pms_per_s": {steps[beta > 0].sum() * n / wall}'
              + (f', "K": {ks}, "phase_s": {{"train": {phase.get("train", 0.0):.3f}, "mutate": {phase.get("mutate", 0.0):.3f}}}' if ks else "")
              + (', "phase_seconds_synchronised": ' + __import__("json").dumps({k: round(v, 4) for k, v in s._core.timing.items()}) if phases else "") + "}", flush=True)
        del s


if __name__ == "__main__":
    main()
