"""G ranks sharing ONE GPU (gloo) against one rank on the bench target: at which iteration, if any, do the two schedules part,
and how far apart are the fitted proposals there?  
    python3 tools/world_parity.py <global particles> <world size>      [HIPCB=1: HIP callbacks instead of torch callbacks]"""
import json, os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def run(n_global, device=0, iters=40):
    import tempest_amd as tp
    from bench import prior20, rosenbrock_torch, ROSENBROCK_HIP
    if os.environ.get("HIPCB") == "1":
        cb = tp.HipCallbacks(ROSENBROCK_HIP, 10)
        prior20, rosenbrock_torch = cb.prior_transform, cb.log_likelihood
    s = tp.Sampler(prior20, rosenbrock_torch, 10, n_particles=n_global, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True, device=device)
    fits = []
    core = s._core
    orig = core.trainer.run
    def spy(weights):
        ms = orig(weights)
        fits.append((ms.means_dev.double().cpu().numpy().ravel().tolist(), ms.chol_dev.double().cpu().numpy().ravel().tolist()))
        return ms
    core.trainer.run = spy
    s.run(n_total=4 * n_global, progress=False)
    st = s.state
    return {"beta": [float(v) for v in st.get_history("beta")], "logz": [float(v) for v in st.get_history("logz")],
            "ess": [float(v) for v in st.get_history("ess")], "steps": [int(v) for v in st.get_history("steps")],
            "acc": [float(v) for v in st.get_history("acceptance")], "fits": fits}

def worker(rank, world, port, n_global, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = run(n_global)
    if rank == 0:
        json.dump(r, open(out, "w"))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    n_global, world = int(sys.argv[1]), int(sys.argv[2])
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    out = f"/tmp/w{world}.json"
    mp.spawn(worker, args=(world, port, n_global, out), nprocs=world, join=True)
    a = json.load(open(out)); b = run(n_global)
    print("iterations", len(a["beta"]), len(b["beta"]))
    for t in range(min(len(a["beta"]), len(b["beta"]))):
        d = {k: abs(a[k][t] - b[k][t]) for k in ("beta", "logz", "ess", "acc")}
        flag = a["steps"][t] != b["steps"][t] or max(d.values()) > 1e-9
        if flag or t < 3:
            print(t, "steps", a["steps"][t], b["steps"][t], {k: (a[k][t], b[k][t]) for k in ("beta", "ess", "acc", "logz")})
        if flag:
            for tt in range(max(0, t - 2), t + 1):
                if tt < len(a["fits"]) and tt < len(b["fits"]):
                    ma, mb = np.array(a["fits"][tt][0]), np.array(b["fits"][tt][0]); ca, cb = np.array(a["fits"][tt][1]), np.array(b["fits"][tt][1])
                    print("fit", tt, "max |dmean|", np.abs(ma - mb).max(), "max |dchol|", np.abs(ca - cb).max(), "mean0", ma[0], mb[0])
            break
    else:
        print("identical schedules; final logz", a["logz"][-1], b["logz"][-1], "bitwise equal logz_t:", a["logz"] == b["logz"], "ess:", a["ess"] == b["ess"], "acc:", a["acc"] == b["acc"])
