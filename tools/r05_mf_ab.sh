#!/bin/bash
# round 5: the screened-batch kernel at 131 072 x 100-D from the prior, round 4's library against this tree's, on ONE box:
# launch times (HIP events), HBM traffic of k_propose_mf (FETCH_SIZE x 64 B... per the guide, WRITE_SIZE), LDS bank conflicts.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/mf_ab; rm -rf $O; mkdir -p $O
ARGS="--d 100 --n 131072 --kernel tpcn --scen prior --reps 7 --variant 6 --epoch 1"
for which in old new; do
  if [ $which = old ]; then L="--older --lib scratch/oldlib/libtempest_hip_r04.so"; else L=""; fi
  timeout -k 10 120 python3 tools/bench_propose.py $ARGS $L > $O/time_$which.jsonl 2> $O/time_$which.err || { echo "timing $which failed"; tail -3 $O/time_$which.err; exit 1; }
  timeout -k 10 120 python3 tools/bench_propose.py --d 50 --n 65536 --kernel tpcn --scen prior --reps 7 --variant 6 --epoch 1 $L >> $O/time_$which.jsonl 2>> $O/time_$which.err
  timeout -k 10 120 python3 tools/bench_propose.py --d 32 --n 262144 --kernel tpcn --scen prior --reps 7 --variant 6 --epoch 1 $L >> $O/time_$which.jsonl 2>> $O/time_$which.err
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/pmc_${which}_$tag -o p -- python3 tools/bench_propose.py $ARGS --reps 3 $L > $O/pmc_${which}_$tag.log 2>&1 || { echo "pmc $tag $which failed"; tail -3 $O/pmc_${which}_$tag.log; exit 1; }
  done
  echo "$which done"; cat $O/time_$which.jsonl | cut -c1-400
done
python3 - <<'PY'
import csv, glob, json, os
O = "gpurun_out/r05/mf_ab"
out = {}
for which in ("old", "new"):
    row = {}
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT"):
        files = glob.glob(f"{O}/pmc_{which}_{tag}/**/*counter_collection.csv", recursive=True)
        acc = {}
        for f in files:
            for r in csv.DictReader(open(f)):
                if "k_propose_mf" not in r.get("Kernel_Name", ""):
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            row[k] = {"launches": len(v), "mean": sum(v) / len(v)}
    out[which] = row
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
rm -rf $O/pmc_*_*/
