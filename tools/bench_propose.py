"""Kernel-level timing of tph_propose (and, with --accept, tph_accept) through the C ABI on synthetic ensembles.

    python tools/bench_propose.py [--lib PATH] [--legacy] [--n 1048576] [--d 10] [--kernel tpcn] [--scen wide,mid,tight]

`--legacy` drives a round-1 library (its tph_propose takes Sigma^-1 where this round's takes L^-1): used to put
before/after numbers of the same box side by side in profiles/.  Prints one JSON line per scenario:
median / min launch duration (HIP events on the launch stream), attempts implied by the in-bounds fraction.
Scenarios mimic the phases of a run on the unit cube: "wide" = proposal as broad as the prior (early iterations,
~50 % of the tpCN proposals out of bounds at d = 10), "mid", "tight" (late iterations, a few %).
"""
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
    ap.add_argument("--lib", default=None)
    ap.add_argument("--legacy", action="store_true")
    ap.add_argument("--older", action="store_true", help="--lib is an earlier round's library (same tph_propose signature, fewer symbols)")
    ap.add_argument("--n", type=int, default=1048576)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--kernel", default="tpcn")
    ap.add_argument("--scen", default="wide,mid,tight")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--accept", action="store_true")
    ap.add_argument("--nocarry", action="store_true")
    ap.add_argument("--wpe", type=int, default=0, help="experiment knob (TPH_OPT_REDRAW_LANES)")
    ap.add_argument("--rounds", type=int, default=0, help="TPH_OPT_BLOCKED (rounds of the blocked kernel; use with --variant 4)")
    ap.add_argument("--scale", type=float, default=0.0, help="overrides the scenario's spread (wide 0.29, mid 0.12, tight 0.04): dials the redraw count")
    ap.add_argument("--sigma-scale", type=float, default=1.0, help="multiplies the step size (RWM runaway: ~8)")
    ap.add_argument("--unstaged", action="store_true", help="TPH_OPT_ML_UNSTAGED = 1 (the redraw-dominated regime of d > 16)")
    ap.add_argument("--lanes", type=int, default=0, help="TPH_OPT_SM_LANES (log2 lanes per particle of the stage-machine kernel)")
    ap.add_argument("--thr", type=int, default=0, help="TPH_OPT_SM_THRESHOLD")
    ap.add_argument("--mflanes", type=int, default=0, help="TPH_OPT_MF_LANES (log2 attempts in flight per particle, screened kernel: --variant 6)")
    ap.add_argument("--audit", action="store_true", help="TPH_OPT_MF_AUDIT = 1")
    ap.add_argument("--noscreen", action="store_true", help="TPH_OPT_SCREEN = 0 (FP64 row walker / multi-lane straggler pass)")
    ap.add_argument("--nomfma", action="store_true", help="TPH_OPT_BLK_MFMA = 0 (blocked rounds through the scalar cache)")
    ap.add_argument("--epoch", type=int, default=0, help="TPH_OPT_MODES_EPOCH (> 0: the packed copies of the factors are built once)")
    ap.add_argument("--pending", type=float, default=0.0,
                    help="fraction of particles with a pending accepted move to resolve (deferred tph_accept), per launch")
    a = ap.parse_args()
    import torch
    from tempest_amd import _lib
    lib = _lib.load(a.lib) if not (a.legacy or a.older) else C.CDLL(a.lib)
    if a.older:                           # an earlier round's library with today's tph_propose signature: bind what it exports
        for name, (res, args) in _lib.SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue
            fn.restype, fn.argtypes = res, args
    if a.legacy:
        for name, (res, args) in _lib.SIGNATURES.items():
            if name in ("tph_fit_modes", "tph_chol_inv", "tph_propose", "tph_accept"):
                continue                      # signatures that changed since round 1
            try:
                fn = getattr(lib, name)
            except AttributeError:            # entry points the round-1 library does not have
                continue
            fn.restype, fn.argtypes = res, args
        lib.tph_propose.restype = C.c_int
        lib.tph_propose.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 6 + [
            C.c_uint64, C.c_uint32, C.c_int64] + [C.c_void_p] * 4
    dev = torch.device("cuda", 0)
    n, d = a.n, a.d
    ctx = C.c_void_p()
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert lib.tph_ctx_create(0, d, 0, C.c_void_p(stream), C.byref(ctx)) == 0
    if a.variant:
        lib.tph_set_option(ctx, 0, a.variant)
    if a.wpe:
        lib.tph_set_option(ctx, 2, a.wpe)
    if a.unstaged:
        lib.tph_set_option(ctx, 3, 1)
    if a.lanes:
        lib.tph_set_option(ctx, 10, a.lanes)
    if a.thr:
        lib.tph_set_option(ctx, 11, a.thr)
    if a.rounds:
        lib.tph_set_option(ctx, 4, a.rounds)
    if a.mflanes:
        lib.tph_set_option(ctx, 13, a.mflanes)
    if a.audit:
        lib.tph_set_option(ctx, 14, 1)
    if a.noscreen:
        lib.tph_set_option(ctx, 12, 0)
    if a.nomfma:
        lib.tph_set_option(ctx, 15, 0)
    if a.epoch:
        lib.tph_set_option(ctx, 5, a.epoch)
    kid = {"tpcn": 0, "rwm": 1}[a.kernel]
    rs = np.random.RandomState(0)
    for scen in a.scen.split(","):
        scale = {"wide": 0.29, "mid": 0.12, "tight": 0.04, "prior": 0.29}[scen]
        if a.scale > 0:
            scale = a.scale
        A = rs.randn(d, d) / np.sqrt(d)
        cov = (A @ A.T + np.eye(d)) * scale ** 2 / 2.0
        L = np.linalg.cholesky(cov)
        u0 = np.clip(0.5 + rs.randn(n, d) @ L.T, 0.001, 0.999) if scen != "prior" else rs.rand(n, d)
        inv = np.linalg.inv(cov)
        W = np.linalg.inv(L)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
        u = t(u0.T)
        means = t(np.full((1, d), 0.5))
        chol = t(L.reshape(1, d, d))
        mat = t((inv if a.legacy else W).reshape(1, d, d))
        dof = t(np.array([1e6]))
        sig = t(np.array([min(2.38 / np.sqrt(d), 0.99) * a.sigma_scale]))
        up = torch.empty_like(u)
        mu_, mup = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
        ctl = torch.zeros(10, dtype=torch.float64, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None   # noqa: E731

        pend = torch.zeros(n, dtype=torch.uint8, device=dev)
        pmask = (torch.rand(n, device=dev) < a.pending).to(torch.uint8)

        def propose(tick, carry):
            args = [ctx, kid, p(u), None, n, n, 1, p(means), p(chol), p(mat), p(dof), p(sig), None, 12345, tick, 0,
                    p(up), p(mu_), p(mup), p(ctl) if carry else None]
            if not a.legacy:
                if a.pending > 0:
                    pend.copy_(pmask)
                args.append(p(pend) if a.pending > 0 else None)
            rc = lib.tph_propose(*args)
            assert rc == 0, lib.tph_last_error()
        propose(1, False)            # fills maha_u
        torch.cuda.synchronize()
        same = float((up == u).all(dim=0).double().mean())
        # share of FIRST attempts that leave the unit cube (host estimate on a subsample, s = 1): what the redraw rounds see
        m = min(n, 100000)
        sg = min(2.38 / np.sqrt(d), 0.99)
        if a.kernel == "tpcn":
            first = 0.5 + np.sqrt(1 - sg * sg) * (u0[:m] - 0.5) + sg * rs.randn(m, d) @ L.T
        else:
            first = u0[:m] + sg * rs.randn(m, d) @ L.T
        inb = float(((first >= 0) & (first <= 1)).all(axis=1).mean())
        ctl[0] = 1.0                 # steps done > 0: the proposal kernels read the carried Mahalanobis form
        ts = []
        for r in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if a.pending > 0:
                pend.copy_(pmask)
                up.copy_(u)          # resolving copies u' over u: keep the ensemble where it is
            e0.record()
            propose(2 + 2 * r, not a.nocarry)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out = {"what": "tph_propose", "lib": os.path.basename(a.lib or "libtempest_hip.so"), "kernel": a.kernel, "n": n, "d": d,
               "scenario": scen, "variant": a.variant, "rounds": a.rounds, "scale": a.scale, "lanes": a.lanes, "z_rows_lds": a.thr, "unstaged": bool(a.unstaged), "nomfma": bool(a.nomfma), "noscreen": bool(a.noscreen), "audit": bool(a.audit), "carry": not a.nocarry, "pending_fraction": a.pending, "median_us": round(float(np.median(ts)), 2),
               "min_us": round(float(np.min(ts)), 2), "first_attempt_in_bounds": inb, "mean_attempts_probe": float(ctl[8].item()), "fell_back_to_current": same,
               "algorithmic_bytes": (16 * d + 4 + 16) * n,
               "GBps_algorithmic": round((16 * d + 20) * n / np.median(ts) / 1e3, 1)}
        if a.variant == 6 or (a.variant == 4 and not a.noscreen):
            cnt = (C.c_ulonglong * 7)()
            assert lib.tph_bench_mf_counters(ctx, cnt) == 0
            out["mf"] = dict(attempts=cnt[1], particles=cnt[2], contradictions=cnt[3], verified=cnt[4], screened=cnt[5], pair_jobs=cnt[6],
                             lanes=a.mflanes, wpe=os.environ.get("TEMPEST_AMD_MF_WPE", "2"))
        print(json.dumps(out), flush=True)
    lib.tph_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
