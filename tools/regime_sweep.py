"""Which d > 16 proposal path is fastest at which redraw rate: the table behind StepEngine._regime (tempest_amd/mcmc.py).

    python tools/regime_sweep.py [--sizes 65536x50,131072x100,262144x32] [--kernel tpcn] [--reps 7]

For every ensemble size and a ladder of proposal spreads (which dial the attempts per particle) it times one tph_propose
launch sequence through: the blocked kernel in R = 1, 2, 3, 4, 6, 8, 12, 16 rounds + the straggler pass, and the screened
batches (propose_mf.hip) with 8 and 16 attempts in flight.  One JSON line per (size, spread): the blocked path's probe
(geometric estimate), the screened path's probe (true mean attempts), the median launch time of every variant."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="65536x50,131072x100,262144x32")
    ap.add_argument("--kernel", default="tpcn")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--scales", default="0.10,0.16,0.19,0.21,0.23,0.25,0.27,0.29")
    ap.add_argument("--rounds", default="1,2,3,4,6,8,12")
    ap.add_argument("--tries", default="1,2,3", help="TPH_OPT_BLK_TRIES values (attempts per round, in place)")
    a = ap.parse_args()
    import torch
    from tempest_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    kid = {"tpcn": 0, "rwm": 1}[a.kernel]
    p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None   # noqa: E731
    for size in a.sizes.split(","):
        n, d = (int(v) for v in size.split("x"))
        rs = np.random.RandomState(0)
        ctx = C.c_void_p()
        stream = torch.cuda.current_stream(dev).cuda_stream
        assert lib.tph_ctx_create(0, d, 0, C.c_void_p(stream), C.byref(ctx)) == 0
        epoch = 0
        for scale in (float(s) for s in a.scales.split(",")):
            A = rs.randn(d, d) / np.sqrt(d)
            cov = (A @ A.T + np.eye(d)) * scale ** 2 / 2.0
            L = np.linalg.cholesky(cov)
            u0 = np.clip(0.5 + rs.randn(n, d) @ L.T, 0.001, 0.999)
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
            u, means, chol, winv = t(u0.T), t(np.full((1, d), 0.5)), t(L.reshape(1, d, d)), t(np.linalg.inv(L).reshape(1, d, d))
            dof, sig = t(np.array([1e6])), t(np.array([min(2.38 / np.sqrt(d), 0.99)]))
            up = torch.empty_like(u)
            mu_, mup = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
            ctl = torch.zeros(10, dtype=torch.float64, device=dev)
            epoch += 1
            lib.tph_set_option(ctx, 5, epoch)          # TPH_OPT_MODES_EPOCH: the packed copies are rebuilt once per ensemble, as in a run

            def launch(tick, carry=True):
                rc = lib.tph_propose(ctx, kid, p(u), None, n, n, 1, p(means), p(chol), p(winv), p(dof), p(sig), None, 12345, tick, 0,
                                     p(up), p(mu_), p(mup), p(ctl) if carry else None, None)
                assert rc == 0, lib.tph_last_error()

            def timed(variant, rounds=0, lanes=0, tries=1):
                lib.tph_set_option(ctx, 0, variant)
                lib.tph_set_option(ctx, 4, rounds)
                lib.tph_set_option(ctx, 13, lanes)
                lib.tph_set_option(ctx, 16, tries)
                ctl[0] = 0.0
                launch(1, False)
                ctl[0] = 1.0
                launch(3)
                ts = []
                for r in range(a.reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch(5 + 2 * r)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                return round(float(np.median(ts)), 1), float(ctl[8].item())
            out = {"n": n, "d": d, "kernel": a.kernel, "scale": scale, "blocked_us": {}, "screened_us": {}}
            for tries in (int(v) for v in a.tries.split(",")):
                for R in (int(v) for v in a.rounds.split(",")):
                    tm, probe = timed(4, R, tries=tries)
                    out["blocked_us"][f"{tries}x{R}"] = tm
                    if tries == 1:
                        out["estimate"] = round(probe, 3)
            for lanes in (3, 4):
                tm, probe = timed(6, 0, lanes)
                out["screened_us"][1 << lanes] = tm
                out["true_mean"] = round(probe, 3)
            best = min(out["blocked_us"], key=out["blocked_us"].get)
            out["best_blocked"] = [best, out["blocked_us"][best]]
            out["best_screened"] = min(out["screened_us"].values())
            print(json.dumps(out), flush=True)
        lib.tph_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
