"""Copy the evidence collected by tools/collect_profiles.sh (gpurun_out/rNN/final) into profiles/ as the round's tracked
summaries:  python3 tools/assemble_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_summary(dirname, kernel_prefix="void k_propose_reg"):
    """Per launch of the proposal kernel: counters summed over the chip, in launch order."""
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        if not r["Kernel_Name"].startswith(kernel_prefix):
            continue
        d = by.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:60], "vgpr": int(r["VGPR_Count"]),
                                                   "duration_us": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def main():
    R = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", R, "final")
    dst = os.path.join(ROOT, "profiles")
    cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))  # noqa: E731
    cp("bench_unprofiled.json", f"{R}_bench_unprofiled.json")
    cp("bench_under_rocprof.json", f"{R}_bench_under_rocprof.json")
    cp(os.path.join("prof_bench", "bench_kernel_stats.csv"), f"{R}_bench_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "bench_trace_summary.json")):
        cp("bench_trace_summary.json", f"{R}_bench_trace_summary.json")
    cp("roofline_table.json", f"{R}_roofline_table.json")
    cp(os.path.join("roof", "meta.json"), f"{R}_roofline_table_meta.json")
    for k in ("rwm", "tpcn"):
        cp(os.path.join(f"prof_c2_{k}", "c2_kernel_stats.csv"), f"{R}_c2_{k}_kernel_stats.csv")
    shutil.copyfile(os.path.join(dst, f"{R}_c2_rwm_kernel_stats.csv"), os.path.join(dst, f"{R}_c2_kernel_stats.csv"))
    c2 = {}
    for k in ("rwm", "tpcn"):
        for tag, name in (("", f"c2_{k}.log"), ("_unprofiled", f"c2_{k}_plain.log")):
            line = [ln for ln in open(os.path.join(src, name)).read().splitlines() if ln.startswith("{")]
            if line:
                c2[k + tag] = json.loads(line[0])            # first run of the process ...
                if len(line) > 1:
                    c2[k + tag + "_second_run_in_process"] = json.loads(line[-1])
    json.dump(c2, open(os.path.join(dst, f"{R}_c2_runs.json"), "w"), indent=1)
    # configs 3 and 5 (shard): kernel stats + the run lines
    others = {}
    for tag, stats, logs in (("c3", os.path.join("prof_c3", "c3_kernel_stats.csv"), (("", "c3.log"), ("_unprofiled", "c3_plain.log"))),
                             ("c5_shard_131072", os.path.join("prof_c5", "c5_kernel_stats.csv"), (("", "c5.log"), ("_unprofiled", "c5_plain.log")))):
        if os.path.exists(os.path.join(src, stats)):
            cp(stats, f"{R}_{tag.split('_')[0]}_kernel_stats.csv")
            for suffix, name in logs:
                if os.path.exists(os.path.join(src, name)):
                    line = [ln for ln in open(os.path.join(src, name)).read().splitlines() if ln.startswith("{")]
                    if line:
                        others[tag + suffix] = json.loads(line[0])
                        if len(line) > 1:
                            others[tag + suffix + "_second_run_in_process"] = json.loads(line[-1])
    for tag, name in (("c5_shard_262144", "c5_262144.log"), ("sep32_262144_K_grows_to_4", "sep_k4.log"), ("sep32_262144_K_1", "sep_k1.log")):
        if os.path.exists(os.path.join(src, name)):
            line = [ln for ln in open(os.path.join(src, name)).read().splitlines() if ln.startswith("{")]
            if line:
                others[tag] = json.loads(line[-1])
    if others:
        json.dump(others, open(os.path.join(dst, f"{R}_c3_c5_runs.json"), "w"), indent=1)
    # shard-size bench lines
    for name in ("bench_131k", "bench_131k_comm"):
        cp(name + ".json", f"{R}_{name}.json")
    if os.path.exists(os.path.join(src, "ab_plain1.json")):
        import statistics
        ab = {"what": "131 072 particles on one GPU, three interleaved pairs on one box: plain run vs the sharded code path forced "
                      "(TEMPEST_AMD_FORCE_COMM=1: real RCCL at world size 1 + the peer-to-peer layer)", "runs": []}
        for i in (1, 2, 3):
            for k in ("plain", "comm"):
                d = json.load(open(os.path.join(src, f"ab_{k}{i}.json")))
                ab["runs"].append({"run": f"{k}{i}", "value": d["value"], "ms_per_step": d["ms_per_step"], "logz": d["logz"],
                                   "tail_phase_seconds": d["mutation_only"]["phase_seconds"]})
        pl = [r["ms_per_step"] for r in ab["runs"] if r["run"].startswith("plain")]
        cm = [r["ms_per_step"] for r in ab["runs"] if r["run"].startswith("comm")]
        ab["median_ms_per_step"] = {"plain": statistics.median(pl), "sharded_path": statistics.median(cm),
                                    "ratio": statistics.median(cm) / statistics.median(pl)}
        json.dump(ab, open(os.path.join(dst, f"{R}_bench_131k_ab.json"), "w"), indent=1)
    if os.path.exists(os.path.join(src, "sd_off1.json")):
        sd = {"what": "bench line (1 048 576 particles) with TEMPEST_AMD_SORTED_DRAWS=0 (one indexed lookup per up-sampling draw) and =1 "
                      "(draws sorted and merged against the cdf: the default), interleaved on one box", "runs": []}
        for i in (1, 2):
            for k in ("off", "on"):
                d = json.load(open(os.path.join(src, f"sd_{k}{i}.json")))
                sd["runs"].append({"run": f"{k}{i}", "value": d["value"], "ms_per_step": d["ms_per_step"], "logz": d["logz"],
                                   "hip_callbacks": d["hip_callbacks"]["value"], "whole_run": d["whole_run"]["value"],
                                   "tail_phase_seconds": d["mutation_only"]["phase_seconds"]})
        json.dump(sd, open(os.path.join(dst, f"{R}_sorted_draws_ab.json"), "w"), indent=1)
    # reweight kernel traffic
    rw = {}
    for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals, durs = [], []
        for f in glob.glob(os.path.join(src, f"pmc_rw_{tag}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("void k_reweight_reduce<1, 8>") and r["Counter_Name"] == ctr:
                    vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
        by = {}
        for disp, v, dur in vals:
            by.setdefault(disp, [0.0, dur])[0] += v
        big = [v for v in by.values() if v[1] > 100.0]                  # the 1.07 GB launches (the small histories take < 40 us)
        rw[ctr] = {"launches": len(big), "mean_counter_KB": sum(v[0] for v in big) / max(1, len(big)),
                   "mean_duration_us": sum(v[1] for v in big) / max(1, len(big))}
    n_rows = 67108864
    read_b = rw["FETCH_SIZE"]["mean_counter_KB"] * 1024 * 2
    write_b = rw["WRITE_SIZE"]["mean_counter_KB"] * 1024
    json.dump({"kernel": "k_reweight_reduce<1, 8>", "n_rows": n_rows, "algorithmic_bytes_per_launch": 16 * n_rows, "counters": rw,
               "correction": "FETCH_SIZE x 1024 B x 2 (gfx950 reports half of a 16-B-per-lane coalesced read stream, "
                             "/opt/skills/guides/MI355X_MICROARCH.md section HBM); WRITE_SIZE x 1024 B as read",
               "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
               "traffic_over_algorithmic": (read_b + write_b) / (16.0 * n_rows),
               "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --roofline-only  "
                          "(and the same with --pmc WRITE_SIZE, separate passes)"},
              open(os.path.join(dst, f"{R}_reweight_pmc.json"), "w"), indent=1)
    # proposal kernel: before / after
    doc = {"what": "tph_propose, tpCN, 1 048 576 particles x 10-D, K = 1, carried Mahalanobis forms; scenarios = share of first "
                   "attempts inside the unit cube (tight 100 %, mid 99.9 %, wide 47-54 %)",
           "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
                      "SQ_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace -- python3 tools/bench_propose.py --scen tight,mid,wide --reps 5 "
                      "[--legacy --lib <round-1 library>]",
           "units": "SQ_* cycle counters in quad-cycles summed over all SIMDs; SQ_INSTS_* wave-instructions per launch"}
    for tag in ("new", "old"):
        d = os.path.join(src, f"pmc_prop_{tag}")
        if not os.path.isdir(d):
            continue
        launches = pmc_summary(d)
        # launches per scenario: 1 (fills maha_u) + 5 timed, three scenarios in order
        per = {}
        for si, scen in enumerate(("tight", "mid", "wide")):
            sel = launches[si * 6 + 1: si * 6 + 6]
            if not sel:
                continue
            keys = [k for k in sel[0] if k.startswith("SQ_")]
            per[scen] = {"launches": len(sel), "kernel": sel[0]["kernel"], "mean_duration_us_profiled": round(sum(x["duration_us"] for x in sel) / len(sel), 2),
                         **{k: round(sum(x[k] for x in sel) / len(sel)) for k in keys}}
            w = per[scen]
            waves = 1048576 / 64
            w["valu_instructions_per_wave64_of_particles"] = round(w["SQ_INSTS_VALU"] / waves, 1)
            w["valu_busy_us_at_2.4GHz"] = round(w["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e3, 1)
        plain = [json.loads(ln) for ln in open(os.path.join(src, f"prop_{tag}_plain.jsonl")) if ln.startswith("{")]
        doc[("this_round" if tag == "new" else "previous_round") if R != "r02" else ("round2" if tag == "new" else "round1")] = {"pmc": per, "unprofiled_median_us": {f'{p["n"]}_{p["scenario"]}': p["median_us"] for p in plain}}
    json.dump(doc, open(os.path.join(dst, f"{R}_propose_pmc.json"), "w"), indent=1)
    # d > 16 proposal kernels
    rows = [json.loads(ln) for ln in open(os.path.join(src, "prop_d50.jsonl")) if ln.startswith("{")]
    json.dump({"what": "tph_propose at d > 16: variant 3 = multi-lane kernel, 4 = blocked kernel (rounds = TPH_OPT_BLOCKED) + straggler "
                       "pass, 5 = row-walker kernel (propose_sm.hip; z_rows_lds = TPH_OPT_SM_THRESHOLD, 0 = 32); scenario 'prior' = an "
                       "ensemble from the prior with a proposal as broad as the prior (the first iterations of a run: tens to hundreds of "
                       "redraw attempts per particle); scale > 0 = the ensemble's spread, dialled to a few attempts per particle (the "
                       "probe of variant 4 is the geometric estimate from the first attempts, that of 3, 5 and 6 the true mean); 6 = screened batches "
                       "(propose_mf.hip; `mf` = its counters: attempts, FP64 verifications, contradictions); variant 4 runs its rounds on the "
                       "FP64 matrix cores and its stragglers through the screened kernel unless the line says nomfma / noscreen; lib = which build", "runs": rows},
              open(os.path.join(dst, f"{R}_propose_d50_d100.json"), "w"), indent=1)
    print("assembled into", dst)


if __name__ == "__main__":
    main()
