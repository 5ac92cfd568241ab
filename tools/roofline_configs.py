"""Per-kernel roofline of the d > 16 configurations ("state the FP64-vector bound per config").

On the GPU box, for cfg in c2 / c3 / c5 (tools/run_config.py: BASELINE configs 2 and 3 at their stated sizes, a 131 072-particle
shard of config 5), five runs of the SAME command, counters in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots):

    rocprofv3 --kernel-trace --stats ...                                 -d OUT/<cfg>/trace
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ...                        -d OUT/<cfg>/fetch
    rocprofv3 --pmc WRITE_SIZE --kernel-trace ...                        -d OUT/<cfg>/write
    rocprofv3 --pmc <PMC_SQ>  --kernel-trace ...                         -d OUT/<cfg>/sq
    rocprofv3 --pmc <PMC_FP>  --kernel-trace ...                         -d OUT/<cfg>/fp

(tools/collect_roofline_configs.sh runs them), then

    python3 tools/roofline_configs.py --collect OUT/<cfg> --config <cfg> --out profiles/r04_roofline_<cfg>.json

which sums every counter per kernel over the whole run and writes, for every kernel with >= 1 % of the run's GPU time:
launches, mean duration, HBM bytes per launch (FETCH_SIZE x 2 -- the gfx950 correction for wide coalesced reads; an upper
bound for indexed access -- plus WRITE_SIZE) and the fraction of the 8 TB/s peak; FP64 FLOP per launch from the instruction
counters ((2 FMA + ADD + MUL) x 64 lanes + 512 per F64 matrix op) and the fraction of the 78.6 TFLOP/s FP64 peak; the share of
SIMD cycles with a vector instruction in flight, the matrix cores' busy share, LDS instructions and bank-conflict share, the
issue-stall and wait shares of the wave cycles -- and the bound those numbers point at.  Durations come from the plain trace;
counter passes serialise dispatches and are used for counts only."""
import argparse
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_SQ = "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY"
PMC_FP = ("SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 "
          "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM")
HBM_PEAK, FP64_PEAK, F16_MFMA_PEAK = 8.0e12, 78.6e12, 2.5e15
N_SIMD, CLOCK = 1024, 2.4e9            # SIMD-cycles per second at the nominal clock (the chip holds less under FP64 load: fractions are lower bounds)

SIZES = {"c2": (65536, 50), "c3": (262144, 32), "c5": (131072, 100)}


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def model(kernel, n, d):
    """Algorithmic bytes / FP64 FLOP per launch of the library's own d > 16 kernels: what the counters are compared with."""
    dp = 16 * ((d + 15) // 16)
    np_ = dp // 16
    mf = 2 * np_ * (np_ + 1) * 2048.0 / 16.0          # FLOP per particle of one triangular product on the matrix cores (padding included)
    if kernel.startswith("k_propose_blkm"):
        tp = kernel.startswith("k_propose_blkm<0")
        return {"algorithmic_bytes": (16.0 * d + 20.0) * n, "model_fp64_flop": n * mf * (2 if tp else 1),
                "note": "full ensemble; a round over a list of failures does a fraction of this (the mean launch mixes both)"}
    if kernel.startswith("k_maha_tile"):
        return {"algorithmic_bytes": (8.0 * d + 16.0) * n, "model_fp64_flop": n * float(d) * d,
                "note": "|L^-1 x|^2 of every particle through the scalar cache (tri.h)"}
    if kernel.startswith("k_propose_mf"):
        return {"algorithmic_bytes": (16.0 * d + 20.0) * n, "model_fp64_flop": None,
                "note": "screen in FP16 / FP32 (counted under mfma_f16 and the FP32 pipes), FP64 only for verified attempts"}
    if kernel.startswith("k_wcov_tiled") or kernel.startswith("k_wcov_mfma"):
        return {"algorithmic_bytes": None, "model_fp64_flop": None, "note": "d^2 FLOP and 8d + 4 B per working-set row"}
    if kernel.startswith("k_gmm_estep"):
        return {"algorithmic_bytes": None, "model_fp64_flop": None, "note": "K d^2 FLOP and 8d B per working-set row"}
    return {}


def read_counters(dirname):
    """kernel -> counter -> sum over the run; kernel -> dispatches."""
    tot, calls = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot.setdefault(k, {})
            tot[k][r["Counter_Name"]] = tot[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                calls[k] = calls.get(k, 0) + 1
    return tot, calls


def collect(src, cfg, out):
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    by = {}
    for r in rows:
        k = short(r["Name"])
        e = by.setdefault(k, {"calls": 0, "total_ns": 0.0})
        e["calls"] += int(r["Calls"])
        e["total_ns"] += float(r["TotalDurationNs"])
    ctr = {}
    for p in ("fetch", "write", "sq", "fp"):
        if os.path.isdir(os.path.join(src, p)):
            ctr[p] = read_counters(os.path.join(src, p))
    n, d = SIZES[cfg]
    table = []
    for k, e in sorted(by.items(), key=lambda kv: -kv[1]["total_ns"]):
        pct = 100.0 * e["total_ns"] / total_ns
        if pct < 1.0:
            continue
        avg_s = e["total_ns"] / e["calls"] * 1e-9
        row = {"kernel": k, "launches": e["calls"], "total_ms": round(e["total_ns"] / 1e6, 2), "percent_of_gpu_time": round(pct, 2),
               "avg_us": round(avg_s * 1e6, 2)}

        def per_launch(p, name):
            if p not in ctr or k not in ctr[p][0] or name not in ctr[p][0][k]:
                return None
            return ctr[p][0][k][name] / max(1, ctr[p][1].get(k, 1))
        fetch, write = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
        if fetch is not None and write is not None:
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
            hb = (2.0 * fetch + write) * 1024.0
            row.update({"hbm_bytes_per_launch": round(hb), "hbm_read_bytes_x2": round(2.0 * fetch * 1024.0), "hbm_write_bytes": round(write * 1024.0),
                        "hbm_GBs": round(hb / avg_s / 1e9, 1), "hbm_frac_of_8TBs": round(hb / avg_s / HBM_PEAK, 4)})
        fma, add, mul = per_launch("fp", "SQ_INSTS_VALU_FMA_F64"), per_launch("fp", "SQ_INSTS_VALU_ADD_F64"), per_launch("fp", "SQ_INSTS_VALU_MUL_F64")
        mops64, mops16 = per_launch("fp", "SQ_INSTS_VALU_MFMA_MOPS_F64"), per_launch("fp", "SQ_INSTS_VALU_MFMA_MOPS_F16")
        if fma is not None:
            valu_flop = (2.0 * fma + (add or 0.0) + (mul or 0.0)) * 64.0
            mfma_flop = (mops64 or 0.0) * 512.0
            row.update({"fp64_flop_per_launch_valu": round(valu_flop), "fp64_flop_per_launch_mfma": round(mfma_flop),
                        "fp64_TFs": round((valu_flop + mfma_flop) / avg_s / 1e12, 3),
                        "fp64_frac_of_78.6TF": round((valu_flop + mfma_flop) / avg_s / FP64_PEAK, 4),
                        "trans_f64_insts": round(per_launch("fp", "SQ_INSTS_VALU_TRANS_F64") or 0.0)})
            if mops16:
                row["mfma_f16_flop_per_launch"] = round(mops16 * 512.0)
                row["mfma_f16_frac_of_2.5PF"] = round(mops16 * 512.0 / avg_s / F16_MFMA_PEAK, 5)
            busy = per_launch("fp", "SQ_VALU_MFMA_BUSY_CYCLES")
            if busy is not None:
                row["mfma_busy_share_of_simd_cycles"] = round(busy / (avg_s * CLOCK * N_SIMD), 4)
            row["smem_insts"] = round(per_launch("fp", "SQ_INSTS_SMEM") or 0.0)
        valu, act = per_launch("sq", "SQ_INSTS_VALU"), per_launch("sq", "SQ_ACTIVE_INST_VALU")
        if valu is not None:
            wc, wi, wa = per_launch("sq", "SQ_WAVE_CYCLES"), per_launch("sq", "SQ_WAIT_INST_ANY"), per_launch("sq", "SQ_WAIT_ANY")
            lds, conf, idx = per_launch("sq", "SQ_INSTS_LDS"), per_launch("sq", "SQ_LDS_BANK_CONFLICT"), per_launch("sq", "SQ_LDS_IDX_ACTIVE")
            row.update({"valu_insts": round(valu), "valu_active_share_of_simd_cycles": round(4.0 * (act or 0.0) / (avg_s * CLOCK * N_SIMD), 4),
                        "lds_insts": round(lds or 0.0), "lds_bank_conflict_share": round((conf or 0.0) / idx, 4) if idx else None,
                        "lds_active_share_of_cu_cycles": round((idx or 0.0) / (avg_s * CLOCK * N_SIMD / 4.0), 4),
                        "issue_stall_share_of_wave_cycles": round((wi or 0.0) / wc, 4) if wc else None,
                        "wait_share_of_wave_cycles": round((wa or 0.0) / wc, 4) if wc else None})
        row.update(model(k, n, d))
        # the bound the numbers point at
        cands = {"hbm": row.get("hbm_frac_of_8TBs") or 0.0, "fp64 (vector + matrix)": row.get("fp64_frac_of_78.6TF") or 0.0,
                 "valu issue": row.get("valu_active_share_of_simd_cycles") or 0.0, "lds": row.get("lds_active_share_of_cu_cycles") or 0.0}
        best = max(cands, key=cands.get)
        if cands[best] >= 0.4:
            row["bound"] = best
        elif (row.get("wait_share_of_wave_cycles") or 0.0) >= 0.5:
            row["bound"] = f"latency: waves parked on memory / barriers {row['wait_share_of_wave_cycles']:.0%} of their cycles (nearest resource: {best} at {cands[best]:.0%})"
        else:
            row["bound"] = f"latency / launch: no resource above 40 % (nearest: {best} at {cands[best]:.0%})"
        row["fractions"] = {k2: round(v, 4) for k2, v in cands.items()}
        table.append(row)
    run_line = None
    log = os.path.join(src, "trace.log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        run_line = json.loads(lines[-1]) if lines else None
    json.dump({"config": cfg, "n_particles": n, "n_dim": d, "gpu_time_ms": round(total_ns / 1e6, 1), "run_under_the_tracer": run_line,
               "peaks": {"hbm_Bps": HBM_PEAK, "fp64_FLOPs": FP64_PEAK, "simd_cycles_per_s_nominal": N_SIMD * CLOCK},
               "how": __doc__.split("\n\n")[0] + " -- tools/roofline_configs.py", "kernels": table}, open(out, "w"), indent=1)
    for r in table:
        print("%-56s %6d x %9.1f us %5.1f %%  hbm %s  fp64 %s  valu %s  -> %s" % (
            r["kernel"][:56], r["launches"], r["avg_us"], r["percent_of_gpu_time"], r.get("hbm_frac_of_8TBs"), r.get("fp64_frac_of_78.6TF"),
            r.get("valu_active_share_of_simd_cycles"), r["bound"][:60]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--collect")
    ap.add_argument("--config", default="c5")
    ap.add_argument("--out")
    ap.add_argument("--print-pmc", choices=("sq", "fp"))
    a = ap.parse_args()
    if a.print_pmc:
        print(PMC_SQ if a.print_pmc == "sq" else PMC_FP)
        return
    collect(a.collect, a.config, a.out)


if __name__ == "__main__":
    sys.exit(main())
