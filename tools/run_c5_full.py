"""BASELINE config 5 at its stated ensemble on ONE MI355X: 100-D Neal funnel, tpCN, 2 097 152 particles, to beta = 1.
One line of JSON: evidence, iterations, MCMC steps, wall time, the weighted mean of v = x_0 over the whole history (formed on the
device: the history is ~10^8 rows), the peak of the device memory in use (sampled after every phase), and where the history's
memory ended up (tph_history_memory: mapped range, growth steps, copies, whether the row-major mirror survived).
Progress lines (one per iteration) go to stderr.  TEMPEST_AMD_RUN_PARTICLES overrides the particle count."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import tempest_amd as tp
    d, n = 100, int(os.environ.get("TEMPEST_AMD_RUN_PARTICLES", "2097152"))
    dev = torch.device("cuda", 0)
    scale = torch.full((d,), 600.0, dtype=torch.float64, device=dev); scale[0] = 30.0
    shift = torch.full((d,), -300.0, dtype=torch.float64, device=dev); shift[0] = -15.0

    def loglike(x):
        v = x[:, 0]
        lv = -0.5 * (v / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)
        lr = (-0.5 * (x[:, 1:] ** 2) * torch.exp(-v)[:, None]).sum(dim=1) - 0.5 * (d - 1) * v - 0.5 * (d - 1) * np.log(2 * np.pi)
        return lv + lr
    total = torch.cuda.mem_get_info(dev)[1]
    peak = {"used": 0}

    def sample_mem():
        free, _ = torch.cuda.mem_get_info(dev)
        peak["used"] = max(peak["used"], total - free)
    s = tp.Sampler(lambda u: u * scale + shift, loglike, d, vectorize=True, n_particles=n, clustering=False, random_state=0,
                   sample="tpcn", backend="torch", batch_prior=True)
    core = s._core
    rows = []
    for name in ("reweighter", "trainer", "resampler", "mutator"):
        obj = getattr(core, name)
        orig = obj.run

        def wrapped(*a, _orig=orig, **k):
            r = _orig(*a, **k)
            sample_mem()
            return r
        obj.run = wrapped
    commit = core.state.commit_current_to_history
    t_last = [time.perf_counter()]

    def commit_logged(*a, **k):
        r = commit(*a, **k)
        torch.cuda.synchronize(dev)
        sample_mem()
        now = time.perf_counter()
        st = core.state
        m = st.ctx.history_memory()
        row = {"iter": st.get_history_length(), "beta": float(st.get_history("beta")[-1]), "steps": int(st.get_history("steps")[-1]),
               "seconds": round(now - t_last[0], 3), "used_gb": round((total - torch.cuda.mem_get_info(dev)[0]) / 2 ** 30, 2),
               "history_rows": m["rows"], "backed_rows": m["rows_backed"], "mirror_rows": m["mirror_rows"], "copies": m["copies"]}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
        t_last[0] = now
        return r
    core.state.commit_current_to_history = commit_logged
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    s.run(n_total=4 * n, progress=False)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    st = s.state
    steps, beta = np.asarray(st.get_history("steps")), np.asarray(st.get_history("beta"))
    ctx = st.ctx
    mtrip = st.reweight_eval([1.0])[0]
    w = ctx.weights(1.0, mtrip[0], mtrip[1])
    mean_u = ctx.weighted_moments(w)[:d].cpu().numpy()
    sample_mem()
    mean_v = float(30.0 * mean_u[0] - 15.0)
    mem = ctx.history_memory()
    out = {"config": "c5_full", "n_dim": d, "n_particles": n, "logz": float(s.evidence()[0]), "analytic_logz": float(-np.log(30.0) - 99 * np.log(600.0)),
           "iterations": int(len(beta)), "mcmc_steps": int(steps[beta > 0].sum()), "wall_s": round(wall, 2),
           "pms_per_s": float(steps[beta > 0].sum() * n / wall), "mean_v": mean_v, "ess_final": float(mtrip[1] ** 2 / mtrip[2]),
           "device_memory_total_gb": round(total / 2 ** 30, 1), "peak_used_gb": round(peak["used"] / 2 ** 30, 1),
           "history_bytes_gb": round(mem["rows"] * (2 * d + 2) * 8 / 2 ** 30, 1), "history_memory": mem,
           "row_mirror": "kept" if mem["mirror_rows"] > 0 else ("given back" if mem["mirror_drops"] else "off"),
           "reference_twin_n4096": "tests/golden/ref_ensembles.json: c5twin_funnel100_n4096", "per_iteration": rows}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
