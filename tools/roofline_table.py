"""Per-kernel roofline table (SURVEY 8d / VERDICT r01 item 6): duration, algorithmic bytes, HBM bytes from counters.

Drive mode (run three times on the GPU box: plain kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE -- separate passes, as
MI355X_MICROARCH.md prescribes):

    rocprofv3 --kernel-trace --output-format csv -d OUT/trace -o t -- python3 tools/roofline_table.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -o f -- python3 tools/roofline_table.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -o w -- python3 tools/roofline_table.py

launches every hot-path kernel a few times on a config-4-sized problem: 10-D, 1 048 576 active particles, a history of
25 committed iterations (26 214 400 rows, 4.6 GB).  Collect mode

    python3 tools/roofline_table.py --collect OUT --out profiles/r02_roofline_table.json

reads the three CSV sets and writes per kernel: launches, mean duration (plain trace), algorithmic bytes per launch (SURVEY
8d's per-unit figure x the units of the launch), achieved GB/s from both, FETCH_SIZE x 2 (gfx950 correction for wide
coalesced reads; indexed-access kernels are marked: for them the x2 is an upper bound) and WRITE_SIZE in bytes, and the
fraction of the 8 TB/s peak.
"""
import argparse
import csv
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

D, N, T = 10, 1048576, 25
REPS = 3
NH = N * T
PEAK = 8000.0

# kernel-name prefix -> (label, algorithmic bytes per launch, units, note)
ALGO = {
    "k_logmix_append": ("K1 log-mixture update at a commit", 24.0 * (NH - N) + 16.0 * N, "24 B per old row + 16 B per new row", "stream"),
    "void k_reweight_reduce<1, 8>": ("K2 reweight reduction, one beta", 16.0 * NH, "16 B per history row", "stream"),
    "void k_reweight_reduce<15, 2>": ("K2 reweight reduction, 15 betas per pass", 16.0 * NH, "16 B per history row (FP64-exp-bound, not HBM)", "stream"),
    "k_weights": ("K3 weights", 24.0 * NH, "24 B per history row", "stream"),
    "void tph_scan::k_tile_sums<0>": ("K6 scan pass 1 (tile sums)", 8.0 * NH, "8 B per row", "stream"),
    "void tph_scan::k_apply<0>": ("K6 scan pass 3 (local scan + offset)", 16.0 * NH, "16 B per row", "stream"),
    "void k_seg_tile_sums<0>": ("K6 scan over the canonical pieces, pass 1 (tile sums)", 8.0 * NH, "8 B per row", "stream"),
    "void k_seg_apply<0>": ("K6 scan over the canonical pieces, pass 3 (local scan + offset)", 16.0 * NH, "16 B per row", "stream"),
    "k_resample_multinomial": ("K6 inverse-CDF lookups of the resampling draws", 8.0 * N, "8 B out per draw (+ ~3 index lines read per draw)", "indexed"),
    "k_multinomial_counts": ("K6 x4 up-sampling, one indexed lookup per draw (TPH_OPT_SORTED_DRAWS = 0)", 0.0, "one atomic per draw (+ ~3 index lines read per draw)", "indexed"),
    "k_mc_draws": ("K6 x4 up-sampling: the draws as 53-bit integers", 0.0, "8 B written per draw", "stream"),
    "k_mc_bounds": ("K6 x4 up-sampling: the stretch of the sorted draws of every tile of the cdf", 0.0, "two bisections of the sorted draws per 2048 rows", "indexed"),
    "k_mc_tiles": ("K6 x4 up-sampling: sorted draws counted by the owners of the cdf's tiles", 0.0, "8 B per draw + 12 B per history row", "stream"),
    "k_gather_rows": ("K7 gather of the resampled rows (from the row-major mirror)", 2.0 * (2 * D + 1) * 8.0 * N, "2 (2d+1) 8 B per output row", "indexed"),
    "k_rows_pack": ("K7 mirror fill: the iteration's new rows, dimension-major -> records", 2.0 * (2 * D + 1) * 8.0 * N, "2 (2d+1) 8 B per new row", "stream"),
    "k_gather(": ("K7 gather of the resampled rows (dimension-major history, no mirror)", 2.0 * (2 * D + 1) * 8.0 * N, "2 (2d+1) 8 B per output row", "indexed"),
    "void k_propose_reg<0, 10, true": ("K9 proposal (tpCN, d = 10)", (16.0 * D + 20.0) * N, "16d + 20 B per particle (VALU-bound)", "stream"),
    "void k_accept<0>": ("K10 Metropolis decision (deferred update)", 49.0 * N, "32 B read + 16 B written + 1 B mask per particle", "stream"),
    "k_adapt": ("K10 sigma adaptation + column sums of the block partials", 16.0 * (N // 256), "16 B per 256 particles", "stream"),
    "void k_wsum<int>": ("K11 first moments of the up-sampled set", (8.0 * D + 4.0) * NH, "8d + 4 B per row (compacted set: fewer rows)", "stream"),
    "void k_wcov_small<int, 10>": ("K11 centred second moments", (8.0 * D + 4.0) * NH, "8d + 4 B per row (compacted set: fewer rows)", "stream"),
    "k_med_hist1": ("K11 median histogram level 1", (8.0 * D + 4.0) * NH, "8d + 4 B per row (compacted set: fewer rows)", "stream"),
    "k_nz_scatter": ("K11 compaction of the rows with multiplicity > 0", 0.0, "8d B read per history row + 8d B written per kept row", "indexed"),
    "void k_wmom_small<10>": ("K4 one-pass weighted moments (volume variation)", (8.0 * D + 8.0) * NH, "8d + 8 B per row", "stream"),
    "void k_cv_sum_small<10>": ("K4 |L^-1 (u - mean)|^2 statistic", (8.0 * D + 8.0) * NH, "8d + 8 B per row", "stream"),
    "void k_membw<0>": ("ceiling: streaming read", 8.0 * 2 * NH, "bytes read", "stream"),
    "void k_membw<1>": ("ceiling: streaming copy", 16.0 * NH, "bytes read + written", "stream"),
}


def drive():
    """A REAL history: the bench workload (10-D Rosenbrock, 1 048 576 particles) run for T iterations, then every hot-path
    kernel launched REPS more times on that state at the trial beta the algorithm would choose next.  Collect mode takes
    the LAST launches of each kernel (the explicit ones)."""
    import torch
    import tempest_amd as tp
    from bench import prior20, rosenbrock_torch
    from tempest_amd.mcmc import PhiloxStream
    dev = torch.device("cuda", 0)
    s = tp.Sampler(prior20, rosenbrock_torch, D, n_particles=N, vectorize=True, clustering=False, random_state=0,
                   backend="torch", batch_prior=True, device=0)
    while s.state.get_history_length() < T:
        s.sample(return_state=False)
    c = s.state.ctx
    assert c.size == NH
    rng = PhiloxStream(11)
    # the trial beta the algorithm itself would pick next: ESS = 2 N (ess_ratio 2), by bisection on the device
    lo_b, hi_b = float(s.state.get_current("beta")), 1.0
    for _ in range(40):
        mid = 0.5 * (lo_b + hi_b)
        m_, s1_, s2_ = c.reweight_eval([mid])[0]
        if s1_ * s1_ / s2_ >= 2 * N:
            lo_b = mid
        else:
            hi_b = mid
    bsel = lo_b
    torch.cuda.synchronize()
    for rep in range(REPS):
        tri = c.reweight_eval([bsel])[0]
        c.reweight_eval(list(np.linspace(bsel, min(1.0, 1.3 * bsel), 15)))
        w = c.weights(bsel, tri[0], tri[1])
        thr = c.trim_threshold(w, 0.99, 1000)
        cdf = c.cdf(w)
        idx = c.resample_multinomial(cdf, N, rng.seed, rng.next())
        uo, xo, lo = c.empty(D, N), c.empty(D, N), c.empty(N)
        c.gather(idx, uo, xo, lo)
        cdfm = c.cdf(w, thr[0:1])
        from tempest_amd.device import OPT_SORTED_DRAWS
        c.set_option(OPT_SORTED_DRAWS, 0)                 # the draw-order form, for the table only
        c.multinomial_counts(cdfm, rng.seed, rng.tick, kept_count=thr[2:3], factor=4, n_draw_max=4 * NH)
        c.set_option(OPT_SORTED_DRAWS, 1)
        counts = c.multinomial_counts(cdfm, rng.seed, rng.next(), kept_count=thr[2:3], factor=4, n_draw_max=4 * NH)
        means, covs, chol, inv, winv = c.fit_modes(counts)
        m_keep = int((counts > 0).sum().item())
        centre = means[0].clone()
        c.volume_variation(w, centre)
        # MCMC steps (deferred update), carried Mahalanobis forms
        from types import SimpleNamespace
        modes = SimpleNamespace(K=1, means_dev=means, chol_dev=chol, winv_dev=winv, dof_dev=torch.tensor([1e6], dtype=torch.float64, device=dev))
        sig = torch.tensor([0.75], dtype=torch.float64, device=dev)
        up, mu_, mup = c.empty(D, N), c.empty(N), c.empty(N)
        ctl = torch.zeros(10, dtype=torch.float64, device=dev)
        ctl[6] = bsel
        pend = torch.zeros(N, dtype=torch.uint8, device=dev)
        part = c.empty(((N + 255) // 256) * 2)
        sums, cnts, sigm = c.zeros(2), torch.tensor([float(N)], dtype=torch.float64, device=dev), sig.clone()
        for step in range(3):
            c.propose("tpcn", uo, None, modes, sigm, None, rng.seed, 1, 0, up, mu_, mup, ctl=ctl, pending=pend)
            lp = rosenbrock_torch(prior20(up.T)).contiguous()
            c.accept("tpcn", bsel, uo, None, lo, up, None, lp, mu_, mup, None, 1, modes.dof_dev, rng.seed, 2, 0, None, ctl=ctl,
                     partials=part, pending=pend)
            c.adapt("tpcn", sums, cnts, 1, N, 1, 20, sigm, ctl, partials=part, n=N)
        c.membw_time(0, 2 * NH, 3)
        c.membw_time(1, NH, 3)
    torch.cuda.synchronize()
    meta = os.environ.get("TPH_ROOFLINE_META")
    if meta:
        json.dump({"rows_with_multiplicity": m_keep, "compacted": bool(2 * m_keep <= NH), "draws": 4 * int(thr[2].item()),
                   "kept_rows": int(thr[2].item()), "beta": bsel, "ess": float(tri[1] ** 2 / tri[2])}, open(meta, "w"))
    print("drive done: history rows", c.size, "rows with multiplicity > 0:", m_keep, "beta", bsel)


def _read(dirname, want_counter=None):
    dur, val = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur.setdefault(r["Kernel_Name"], []).append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    if want_counter:
        for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == want_counter:
                    val.setdefault((r["Kernel_Name"], int(r["Dispatch_Id"])), 0.0)
                    val[(r["Kernel_Name"], int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
    per = {}
    for (k, disp), v in sorted(val.items(), key=lambda kv: kv[0][1]):
        per.setdefault(k, []).append(v)
    dur = {k: [d for _, d in sorted(v)] for k, v in dur.items()}      # launch order
    return dur, per


def collect(base, out):
    meta = json.load(open(os.path.join(base, "meta.json")))
    rows_fit = meta["rows_with_multiplicity"] if meta["compacted"] else NH
    for key in ("void k_wsum<int>", "void k_wcov_small<int, 10>", "k_med_hist1"):
        label, _, unit, kind = ALGO[key]
        ALGO[key] = (label, (8.0 * D + 4.0) * rows_fit, f"8d + 4 B per row of the working set ({rows_fit} rows with multiplicity > 0)", kind)
    label, _, unit, kind = ALGO["k_nz_scatter"]
    ALGO["k_nz_scatter"] = (label, 4.0 * NH + (16.0 * D + 4.0) * rows_fit, "4 B per history row + (16d + 4) B per kept row", kind)
    label, _, unit, kind = ALGO["k_multinomial_counts"]
    ALGO["k_multinomial_counts"] = (label, 0.0, f"{meta['draws']} draws: one atomic each (+ ~3 index lines read per draw)", kind)
    label, _, unit, kind = ALGO["k_mc_draws"]
    ALGO["k_mc_draws"] = (label, 8.0 * meta["draws"], f"8 B written per draw ({meta['draws']} draws)", kind)
    label, _, unit, kind = ALGO["k_mc_tiles"]
    ALGO["k_mc_tiles"] = (label, 8.0 * meta["draws"] + 12.0 * NH, f"8 B per draw ({meta['draws']}) + 12 B per history row (8 B of cdf read, 4 B of "
                          "count written)", kind)
    dur, _ = _read(os.path.join(base, "trace"))
    _, fetch = _read(os.path.join(base, "fetch"), "FETCH_SIZE")
    _, write = _read(os.path.join(base, "write"), "WRITE_SIZE")
    rows = []
    for prefix, (label, algo, unit, kind) in ALGO.items():
        names = [k for k in dur if k.startswith(prefix)]
        if not names:
            continue
        name = max(names, key=lambda k: np.mean(dur[k]))
        # the explicit launches of the drive come last: REPS repetitions x `per` launches each
        per = {"void k_propose_reg": 3, "void k_accept": 3, "k_adapt": 3, "void k_membw": 5, "void tph_scan::k_tile_sums<0>": 1,
               "void tph_scan::k_apply<0>": 1}
        k_last = REPS * next((v for p_, v in per.items() if prefix.startswith(p_)), 1)
        big = np.asarray(dur[name][-k_last:])
        t_us = float(np.mean(big)) / 1e3
        fz = np.asarray(fetch.get(name, [0.0])[-k_last:])
        wz = np.asarray(write.get(name, [0.0])[-k_last:])
        f_b = float(np.mean(fz)) * 1024.0 * 2.0
        w_b = float(np.mean(wz)) * 1024.0
        rec = {"kernel": name[:120], "what": label, "launches_measured": int(len(big)), "mean_duration_us": round(t_us, 2),
               "algorithmic_bytes_per_launch": algo, "algorithmic_unit": unit,
               "achieved_GBs_algorithmic": round(algo / t_us / 1e3, 1) if algo else None,
               "frac_of_8TBs": round(algo / t_us / 1e3 / PEAK, 4) if algo else None,
               "FETCH_SIZE_bytes_x2": f_b, "WRITE_SIZE_bytes": w_b,
               "counter_bytes_per_launch": f_b + w_b, "achieved_GBs_counters": round((f_b + w_b) / t_us / 1e3, 1),
               "traffic_over_algorithmic": round((f_b + w_b) / algo, 3) if algo else None,
               # indexed kernels: the counter tallies REQUESTS x 64 B and a lone 64-byte sector is one request (profiles/
               # r05_fetch_calibration.json) -- their read bytes lie between the raw counter and twice it
               "traffic_over_algorithmic_min": round((f_b / 2.0 + w_b) / algo, 3) if algo and kind == "indexed" else None,
               "read_requests_per_us": round(f_b / 2.0 / 64.0 / t_us, 1),
               "access": kind}
        rows.append(rec)
    doc = {"problem": {"n_dim": D, "active_particles": N, "history_rows": NH, "iterations": T},
           "peak_GBs": PEAK,
           "how": "rocprofv3: plain --kernel-trace for durations, --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE "
                  "KB x 1024 x 2 (gfx950: the counter is read requests x 64 B and a request for both 64-byte sectors of a 128-byte line is "
                  "tallied once -- every coalesced stream, 4 / 8 / 16 B per lane, reads exactly half its bytes; a lone sector reads "
                  "exactly: profiles/r05_fetch_calibration.json; for 'indexed' kernels the x2 is an upper bound and "
                  "traffic_over_algorithmic_min the lower one; ~48 requests/ns is the most the probe's gathers reach), WRITE_SIZE KB x 1024",
           "kernels": rows}
    json.dump(doc, open(out, "w"), indent=1)
    for r in rows:
        print(f'{r["what"][:52]:52s} {r["mean_duration_us"]:9.1f} us  algo {r["achieved_GBs_algorithmic"]}  counters {r["achieved_GBs_counters"]} GB/s  '
              f'traffic/algo {r["traffic_over_algorithmic"]}')


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--collect", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_roofline_table.json"))
    a = ap.parse_args()
    if a.collect:
        collect(a.collect, a.out)
    else:
        drive()
