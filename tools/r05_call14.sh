#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call14; rm -rf $O; mkdir -p $O
for cfg in "c2 tpcn" "c3 tpcn"; do
  set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$1 -o s -- python3 tools/run_config.py $1 $2 > $O/run_$1.json 2> $O/run_$1.err || { echo "run failed"; exit 1; }
  find $O/t_$1 -name "*kernel_trace.csv" -delete
done
echo ok
