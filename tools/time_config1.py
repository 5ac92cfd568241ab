"""BASELINE config 1 (the reference's README example: 10-D Rosenbrock, 1 000 particles, default settings incl. clustering) on
one GPU, three seeds, with the step replayed from a graph (the default at this size) and launched step by step; wall time of
Sampler(...) + run()."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main():
    import torch
    import tempest_amd as tp

    def prior(u):
        return 20 * u - 10

    def like(x):
        return -(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2).sum(dim=1)
    for rep in range(3):
        for graph in (None, False):
            t0 = time.perf_counter()
            s = tp.Sampler(prior, like, 10, n_particles=1000, vectorize=True, random_state=rep, graph=graph)
            s.run(progress=False)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            steps = np.asarray(s.state.get_history("steps"))
            print(f'{{"config": "c1", "seed": {rep}, "graph": {"null" if graph is None else "false"}, "wall_s": {wall:.3f}, '
                  f'"iterations": {len(steps)}, "mcmc_steps": {int(steps.sum())}, "logz": {s.evidence()[0]:.4f}}}')


if __name__ == "__main__":
    main()
