#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/blkm_stage; rm -rf $O; mkdir -p $O
timeout -k 10 240 python3 -m pytest tests/test_kernels_gpu.py -q -x -k "staged_in_lds" > $O/test.log 2>&1
rc=$?; echo "staged == streamed test rc=$rc"; tail -4 $O/test.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
for stage in 0 1 0 1; do
  export TEMPEST_AMD_BLK_STAGE=$stage
  for cfg in "100 131072 tight" "100 131072 mid" "50 65536 tight" "50 65536 mid"; do
    set -- $cfg
    timeout -k 10 100 python3 tools/bench_propose.py --d $1 --n $2 --kernel tpcn --scen $3 --reps 11 --variant 4 --rounds 1 --epoch 1 >> $O/time_stage$stage.jsonl 2>> $O/err.log || { echo "bench failed"; tail -3 $O/err.log; exit 1; }
  done
done
python3 - <<'PY'
import json
for tag in ("stage0","stage1"):
    rows=[json.loads(l) for l in open(f"gpurun_out/r05/blkm_stage/time_{tag}.jsonl")]
    print(tag, [(r["d"], r["scenario"], r["median_us"], r["min_us"]) for r in rows])
PY
