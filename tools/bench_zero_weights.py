"""Volume-variation call (tph_volume_variation: k_wmom_small + k_cv_sum_small at d = 10) on a 3.4e7-row history whose first
0 / 25 / 50 / 75 % of rows have weight exactly zero: what skipping weightless rows buys.

    python3 tools/bench_zero_weights.py
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tempest_amd.device import HipContext
d, n = 10, 32 * 1048576
dev = torch.device("cuda", 0)
c = HipContext(d, 0, n)
rs = np.random.RandomState(0)
blk = 1 << 20
for t in range(n // blk):
    u = torch.rand(d, blk, dtype=torch.float64, device=dev)
    c.history_append(u, u, torch.zeros(blk, dtype=torch.float64, device=dev), 0.1 * t, 0.0, blk)
for frac in (0.0, 0.25, 0.5, 0.75):
    w = torch.rand(n, dtype=torch.float64, device=dev)
    w[: int(frac * n)] = 0.0
    w /= w.sum()
    centre = torch.full((d,), 0.5, dtype=torch.float64, device=dev)
    v = c.volume_variation(w, centre)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); v = c.volume_variation(w, centre); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(frac, v, round(min(ts) * 1e6, 1), "us", flush=True)
