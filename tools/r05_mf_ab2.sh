#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/mf_ab2; rm -rf $O; mkdir -p $O
for deal in 0 1; do
  export TEMPEST_AMD_MF_DEAL=$deal
  for cfg in "100 131072" "50 65536" "32 262144"; do
    set -- $cfg
    timeout -k 10 120 python3 tools/bench_propose.py --d $1 --n $2 --kernel tpcn --scen prior --reps 9 --variant 6 --epoch 1 >> $O/time_deal$deal.jsonl 2>> $O/err.log
  done
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$deal -o p -- python3 tools/bench_propose.py --d 100 --n 131072 --kernel tpcn --scen prior --reps 3 --variant 6 --epoch 1 > $O/pmc_$deal.log 2>&1
  python3 - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("$O/pmc_$deal/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_propose_mf" in r.get("Kernel_Name","") and r["Counter_Name"]=="FETCH_SIZE"]
print("deal $deal FETCH_SIZE mean", sum(v)/max(1,len(v)), len(v))
PY
  rm -rf $O/pmc_$deal
done
for old in 1; do
  unset TEMPEST_AMD_MF_DEAL
  for cfg in "100 131072" "50 65536" "32 262144"; do
    set -- $cfg
    timeout -k 10 120 python3 tools/bench_propose.py --d $1 --n $2 --kernel tpcn --scen prior --reps 9 --variant 6 --epoch 1 --older --lib scratch/oldlib/libtempest_hip_r04.so >> $O/time_old.jsonl 2>> $O/err.log
  done
done
python3 - <<'PY'
import json
for tag in ("old","deal0","deal1"):
    rows=[json.loads(l) for l in open(f"gpurun_out/r05/mf_ab2/time_{tag}.jsonl")]
    print(tag, [(r["d"], r["median_us"], r["min_us"]) for r in rows])
PY
