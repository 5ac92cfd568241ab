#!/bin/bash
# Regenerates the round's evidence under gpurun_out/rNN/final on the GPU box (then copied into profiles/ by
# tools/assemble_profiles.py).  Usage: gpurun -- 'bash tools/collect_profiles.sh r04 [A|B]'  (A: sections 1-5, B: sections 6-8)
set -e
R=${1:-r04}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
PART=${2:-all}
O=gpurun_out/$R/final
if [ "$PART" != B ]; then rm -rf "$O"; fi
mkdir -p "$O"
if [ "$PART" != B ]; then
# 1. the bench line, plain
python3 bench.py > "$O/bench_unprofiled.json" 2> "$O/bench_unprofiled.err"
echo "bench done"
# 2. the same under the kernel tracer
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bench" -o bench -- python3 bench.py --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err"
echo "bench under rocprof done"
# 3. reweight kernel: HBM traffic counters, separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_rw_fetch" -o f -- python3 bench.py --roofline-only > "$O/rw_fetch.json" 2> "$O/rw_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_rw_write" -o w -- python3 bench.py --roofline-only > "$O/rw_write.json" 2> "$O/rw_write.err"
echo "pmc K2 done"
# 4. proposal kernel (d <= 16): instruction counters, this round's library and round 2's (scratch/oldlib, if present) on this box
PMC="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$O/pmc_prop_new" -o p -- python3 tools/bench_propose.py --scen tight,mid,wide --reps 5 > "$O/prop_new.jsonl" 2> "$O/prop_new.err"
python3 tools/bench_propose.py --scen tight,mid,wide > "$O/prop_new_plain.jsonl" 2>> "$O/prop_new.err"
python3 tools/bench_propose.py --scen tight,mid,wide --n 131072 >> "$O/prop_new_plain.jsonl" 2>> "$O/prop_new.err"
if [ -f scratch/oldlib/libtempest_hip_r02.so ]; then
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$O/pmc_prop_old" -o p -- python3 tools/bench_propose.py --older --lib scratch/oldlib/libtempest_hip_r02.so --scen tight,mid,wide --reps 5 > "$O/prop_old.jsonl" 2> "$O/prop_old.err"
  python3 tools/bench_propose.py --older --lib scratch/oldlib/libtempest_hip_r02.so --scen tight,mid,wide > "$O/prop_old_plain.jsonl" 2>> "$O/prop_old.err"
  python3 tools/bench_propose.py --older --lib scratch/oldlib/libtempest_hip_r02.so --scen tight,mid,wide --n 131072 >> "$O/prop_old_plain.jsonl" 2>> "$O/prop_old.err"
fi
echo "propose d10 done"
# 5. d > 16 proposal kernels: multi-lane (3), blocked + stragglers (4), row walker (5); "prior" = the first iterations of a run
for k in rwm tpcn; do
  python3 tools/bench_propose.py --d 50 --n 65536 --kernel $k --scen prior --reps 7 --variant 3 --unstaged >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d 50 --n 65536 --kernel $k --scen prior --reps 7 --variant 5 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d 50 --n 65536 --kernel $k --scen tight,mid,wide --reps 10 --variant 3 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d 50 --n 65536 --kernel $k --scen wide --reps 10 --variant 5 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d 50 --n 65536 --kernel $k --scen tight,mid --reps 10 --variant 4 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
done
python3 tools/bench_propose.py --d 50 --n 65536 --kernel rwm --scen prior --reps 5 --variant 3 --unstaged --sigma-scale 4 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 50 --n 65536 --kernel rwm --scen prior --reps 5 --variant 5 --sigma-scale 4 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 32 --n 262144 --kernel tpcn --scen prior --reps 5 --variant 3 --unstaged >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 32 --n 262144 --kernel tpcn --scen prior --reps 5 --variant 5 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 100 --n 262144 --kernel tpcn --scen tight --reps 10 --variant 3 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 100 --n 262144 --kernel tpcn --scen tight --reps 10 --variant 4 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
# 100-D (config 5's shard): multi-lane kernel vs row walker from the prior; rows of z kept in LDS
for sc in prior wide; do
  python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen $sc --reps 5 --variant 3 --unstaged >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen $sc --reps 5 --variant 5 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
done
for zl in 16 48 64; do
  python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen prior --reps 5 --variant 5 --thr $zl >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
done
# steps of a few attempts per particle: multi-lane kernel, blocked kernel in 1 / R rounds + stragglers, row walker (the regime table of DESIGN 3b)
mid() {   # n d scale rounds
  python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen mid --scale $3 --reps 6 --variant 3 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen mid --scale $3 --reps 6 --variant 4 --rounds 1 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen mid --scale $3 --reps 6 --variant 4 --rounds $4 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen mid --scale $3 --reps 6 --variant 5 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
}
mid 65536 50 0.22 2; mid 65536 50 0.25 6; mid 262144 32 0.24 6; mid 262144 32 0.28 10; mid 131072 100 0.21 10; mid 131072 100 0.23 10
# round 4: screened batches (variant 6, propose_mf.hip) from the prior and at a few attempts per particle; blocked rounds on the FP64
# matrix cores (variant 4, the default) against the scalar-cache rounds (--nomfma) and against the FP64 straggler pass (--noscreen)
for size in "65536 50" "262144 32" "131072 100"; do
  set -- $size
  for sc in prior wide; do
    python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen $sc --reps 7 --variant 6 --epoch 1 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  done
  python3 tools/bench_propose.py --d $2 --n $1 --kernel rwm --scen prior --reps 7 --variant 6 --epoch 1 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  for sc in tight mid; do
    python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen $sc --reps 10 --variant 4 --rounds 1 --epoch 1 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
    python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen $sc --reps 10 --variant 4 --rounds 1 --epoch 1 --nomfma >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
    python3 tools/bench_propose.py --d $2 --n $1 --kernel tpcn --scen $sc --reps 10 --variant 4 --rounds 1 --epoch 1 --noscreen >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  done
done
python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen mid --scale 0.23 --reps 7 --variant 6 --epoch 1 >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen prior --reps 5 --variant 6 --epoch 1 --audit >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
if [ -f scratch/oldlib/libtempest_hip_r02.so ]; then      # the multi-lane kernel of round 2 on this box (before the shorter RNG chain)
  python3 tools/bench_propose.py --older --lib scratch/oldlib/libtempest_hip_r02.so --d 50 --n 65536 --kernel rwm --scen prior --reps 7 --variant 3 --unstaged >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
  python3 tools/bench_propose.py --older --lib scratch/oldlib/libtempest_hip_r02.so --d 50 --n 65536 --kernel tpcn --scen prior --reps 7 --variant 3 --unstaged >> "$O/prop_d50.jsonl" 2>> "$O/prop_d50.err"
fi
echo "propose d50 done"
# per-launch summary of the roofline kernel and the proposal kernel out of the bench trace (the raw trace is dropped below)
python3 - "$O" <<'PY'
import csv, json, sys
o = sys.argv[1]
k2, prop = [], []
for r in csv.DictReader(open(o + "/prof_bench/bench_kernel_trace.csv")):
    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    if r["Kernel_Name"].startswith("void k_reweight_reduce<1, 8>") and dur > 140.0:      # the 1.07 GB history only
        k2.append(dur)
    elif r["Kernel_Name"].startswith("void k_propose_reg<0, 10, true"):
        prop.append(dur)
json.dump({"k_reweight_reduce<1, 8> on the 1.07 GB history": {"launches": len(k2), "mean_us": sum(k2) / max(1, len(k2)), "min_us": min(k2), "max_us": max(k2)},
           "k_propose_reg<0, 10, true, *, false>": {"launches": len(prop), "mean_us": sum(prop) / max(1, len(prop))}},
          open(o + "/bench_trace_summary.json", "w"), indent=1)
PY
fi
if [ "$PART" = A ]; then
find "$O" -name "*kernel_trace.csv" -size +8M -delete
find "$O" -name "*counter_collection.csv" -size +20M -delete
du -sh "$O"; echo "collected part A"; exit 0; fi
# 6. configs 2, 3 and a config-5 shard end to end under the tracer
for k in rwm tpcn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2_$k" -o c2 -- python3 tools/run_config.py c2 $k > "$O/c2_$k.log" 2>&1
done
# (unprofiled: the run twice in one process -- the first pays the process's one-off costs, the second is what a long job sees)
TEMPEST_AMD_RUN_REPEAT=2 python3 tools/run_config.py c2 rwm > "$O/c2_rwm_plain.log" 2>&1
TEMPEST_AMD_RUN_REPEAT=2 python3 tools/run_config.py c2 tpcn > "$O/c2_tpcn_plain.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c3" -o c3 -- python3 tools/run_config.py c3 tpcn > "$O/c3.log" 2>&1
TEMPEST_AMD_RUN_REPEAT=2 python3 tools/run_config.py c3 tpcn > "$O/c3_plain.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c5" -o c5 -- python3 tools/run_config.py c5 tpcn > "$O/c5.log" 2>&1
TEMPEST_AMD_RUN_REPEAT=2 python3 tools/run_config.py c5 tpcn > "$O/c5_plain.log" 2>&1
# config 5's shard at its 8-GPU size (2 097 152 / 8), and the separable 32-D four-mode target with K growing to 4 / held at 1
TEMPEST_AMD_RUN_PARTICLES=262144 python3 tools/run_config.py c5 tpcn > "$O/c5_262144.log" 2>&1
TEMPEST_AMD_RUN_MAX_POINTS=4096 python3 tools/run_config.py sep tpcn > "$O/sep_k4.log" 2>&1
python3 tools/run_config.py sep1 tpcn > "$O/sep_k1.log" 2>&1
echo "configs done"
# 7. per-kernel roofline table
mkdir -p "$O/roof"
TPH_ROOFLINE_META="$O/roof/meta.json" python3 tools/roofline_table.py > "$O/roof/plain.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$O/roof/trace" -o t -- python3 tools/roofline_table.py > "$O/roof/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/roof/fetch" -o f -- python3 tools/roofline_table.py > "$O/roof/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/roof/write" -o w -- python3 tools/roofline_table.py > "$O/roof/write.log" 2>&1
python3 tools/roofline_table.py --collect "$O/roof" --out "$O/roofline_table.json" > "$O/roofline_table.txt"
rm -rf "$O/roof/trace" "$O/roof/fetch" "$O/roof/write"
echo "roofline table done"
# 8. config 4's 8-GPU shard size, un-sharded and through the sharded code path over RCCL at world size 1
python3 bench.py --particles 131072 --no-roofline --no-cpu-baseline > "$O/bench_131k.json" 2> "$O/bench_131k.err"
TEMPEST_AMD_FORCE_COMM=1 python3 bench.py --particles 131072 --no-roofline --no-cpu-baseline > "$O/bench_131k_comm.json" 2> "$O/bench_131k_comm.err"
for i in 1 2 3; do
  python3 bench.py --particles 131072 --no-roofline --no-cpu-baseline --no-hip-callbacks > "$O/ab_plain$i.json" 2>> "$O/ab.err"
  TEMPEST_AMD_FORCE_COMM=1 python3 bench.py --particles 131072 --no-roofline --no-cpu-baseline --no-hip-callbacks > "$O/ab_comm$i.json" 2>> "$O/ab.err"
done
# keep the merge small: drop the raw traces that are not summarised further
find "$O" -name "*kernel_trace.csv" -size +8M -delete
find "$O" -name "*counter_collection.csv" -size +20M -delete
du -sh "$O"
echo collected
