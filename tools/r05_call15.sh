#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call15; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "volume or moment or cov or fit or reweight" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E " $O/tests.log | head; exit $rc; fi
for cfg in "c2 tpcn" "c3 tpcn" "c5 tpcn"; do
  set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$1 -o s -- python3 tools/run_config.py $1 $2 > $O/run_$1.json 2> $O/run_$1.err || { echo "run failed"; exit 1; }
  find $O/t_$1 -name "*kernel_trace.csv" -delete
done
bash tools/r05_ab_r04.sh
