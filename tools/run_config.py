"""One BASELINE.json configuration end to end on one GPU (used under rocprofv3 for profiles/): `run_config.py c2|c3|c5 rwm|tpcn`
(c5: a 131 072-particle shard of config 5's 100-D funnel)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main():
    import torch
    import tempest_amd as tp
    which, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "tpcn")
    dev = torch.device("cuda", 0)
    assert which in ("c2", "c3", "c5")
    if which == "c3":       # 32-D four-mode Gaussian mixture, 262 144 particles, clustering (BASELINE config 3)
        d, n = 32, 262144
        mus = torch.zeros(4, d, dtype=torch.float64, device=dev)
        for k, (a, b) in enumerate([(-4, -4), (-4, 4), (4, -4), (4, 4)]):
            mus[k, 0], mus[k, 1] = a, b
        const = float(-np.log(4.0) - 0.5 * d * np.log(2 * np.pi * 0.25))

        def loglike(x):
            q = ((x[:, None, :] - mus[None]) ** 2).sum(dim=2)
            return torch.logsumexp(-0.5 * q / 0.25, dim=1) + const
        s = tp.Sampler(lambda u: 20 * u - 10, loglike, d, vectorize=True, n_particles=n, clustering=True, random_state=0,
                       sample=kernel, backend="torch", batch_prior=True)
    elif which == "c5":     # 100-D Neal funnel, a 131 072-particle shard of BASELINE config 5's 2 097 152
        d, n = 100, 131072
        scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
        shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

        def loglike(x):
            v = x[:, 0]
            lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
            lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
            return lv + lr
        s = tp.Sampler(lambda u: u * scale + shift, loglike, d, vectorize=True, n_particles=n, clustering=False, random_state=0,
                       sample=kernel, backend="torch", batch_prior=True)
    else:
        d, n = 50, 65536
        A = np.random.RandomState(1).randn(d, d)
        S = A @ A.T / d + 0.5 * np.eye(d)
        P = torch.from_numpy(np.linalg.inv(S)).to(dev)
        const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
        s = tp.Sampler(lambda u: 20 * u - 10, lambda x: -0.5 * ((x @ P) * x).sum(dim=1) + const, d, vectorize=True,
                       n_particles=n, clustering=False, random_state=0, sample=kernel, backend="torch", batch_prior=True)
    t0 = time.perf_counter()
    s.run(n_total=4 * n, progress=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    steps = np.asarray(s.state.get_history("steps")); beta = np.asarray(s.state.get_history("beta"))
    print(f'{{"config": "{which}", "kernel": "{kernel}", "n_dim": {d}, "n_particles": {n}, "logz": {s.evidence()[0]}, '
          f'"analytic_logz": {-d * np.log(20.0) if which != "c5" else -np.log(30.0) - 99 * np.log(600.0)}, "iterations": {len(beta)}, "mcmc_steps": {int(steps[beta > 0].sum())}, '
          f'"wall_s": {wall}, "pms_per_s": {steps[beta > 0].sum() * n / wall}}}')


if __name__ == "__main__":
    main()
