"""BASELINE config 2 (50-D Gaussian, 65 536 particles): redraw attempts per particle, sigma / sigma_0 and accepted fraction at the
last MCMC step of every iteration -- the log behind DESIGN.md section 9 (RWM step-size runaway at small beta).
    python3 tools/config2_attempts.py rwm|tpcn"""
import sys, json, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
import tempest_amd as tp
from tempest_amd import mcmc
kernel = sys.argv[1] if len(sys.argv) > 1 else "rwm"
d, n = 50, 65536
dev = torch.device("cuda", 0)
A = np.random.RandomState(1).randn(d, d)
S = A @ A.T / d + 0.5 * np.eye(d)
P = torch.from_numpy(np.linalg.inv(S)).to(dev)
const = float(-0.5 * np.linalg.slogdet(S)[1] - 0.5 * d * np.log(2 * np.pi))
def prior(u): return 20 * u - 10
def like(x): return -0.5 * ((x @ P) * x).sum(dim=1) + const
log = []
orig = mcmc.StepEngine.wait_record
def spy(self, step, timeout=None):
    rec = orig(self, step, timeout)
    log.append((float(self.mailbox_np[step % self.SLOTS][6]), bool(self.unstaged), bool(self.blocked), float(rec[4]), float(rec[2])))
    return rec
mcmc.StepEngine.wait_record = spy
s = tp.Sampler(prior, like, d, vectorize=True, n_particles=n, clustering=False, random_state=0, sample=kernel, backend="torch", batch_prior=True)
t0 = time.perf_counter(); s.run(n_total=4 * n, progress=False); torch.cuda.synchronize(); wall = time.perf_counter() - t0
a = np.array([x[0] for x in log]); un = np.array([x[1] for x in log]); bl = np.array([x[2] for x in log])
steps = s.state.get_history("steps")
print(json.dumps({"kernel": kernel, "wall_s": wall, "records": len(a), "steps_total": int(np.sum(steps)),
                  "mean_attempts_overall": float(a.mean()), "sum_attempts_recorded": float(a.sum()),
                  "unstaged_records": int(un.sum()), "mean_attempts_unstaged": float(a[un].mean()) if un.any() else None,
                  "blocked_records": int(bl.sum()), "quantiles_unstaged": [float(q) for q in np.quantile(a[un], [0, .1, .5, .9, 1])] if un.any() else None,
                  "per_iteration(attempts, sigma/sigma0, accepted)": [(round(x[0], 1), round(x[3], 2), round(x[4], 3)) for x in log]}))
