#!/bin/bash
# Per-kernel roofline of configs 2, 3 and the config-5 shard (tools/roofline_configs.py): five profiled runs per config on the GPU
# box, counters in passes of their own; the per-dispatch CSVs stay on the box, only the per-kernel summaries come back.
# Usage: gpurun -- 'bash tools/collect_roofline_configs.sh r04 [c2 c3 c5]'
R=${1:-r04}; shift
CFGS=${@:-c2 c3 c5}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${R}_roofline; mkdir -p $O
W=/tmp/roofline_$$; rm -rf $W; mkdir -p $W
SQ=$(python3 tools/roofline_configs.py --print-pmc sq); FP=$(python3 tools/roofline_configs.py --print-pmc fp)
for cfg in $CFGS; do
  k=tpcn
  mkdir -p $W/$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W/$cfg/trace -o t -- python3 tools/run_config.py $cfg $k > $W/$cfg/trace.log 2>&1 || { echo "trace failed: $cfg"; tail -3 $W/$cfg/trace.log; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $W/$cfg/fetch -o f -- python3 tools/run_config.py $cfg $k > $W/$cfg/fetch.log 2>&1 || { echo "fetch failed: $cfg"; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $W/$cfg/write -o w -- python3 tools/run_config.py $cfg $k > $W/$cfg/write.log 2>&1 || { echo "write failed: $cfg"; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $W/$cfg/sq -o s -- python3 tools/run_config.py $cfg $k > $W/$cfg/sq.log 2>&1 || { echo "sq failed: $cfg"; tail -3 $W/$cfg/sq.log; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc $FP --kernel-trace --output-format csv -d $W/$cfg/fp -o p -- python3 tools/run_config.py $cfg $k > $W/$cfg/fp.log 2>&1 || { echo "fp failed: $cfg"; tail -3 $W/$cfg/fp.log; exit 1; }
  echo "== $cfg"; grep -h "^{" $W/$cfg/trace.log | cut -c1-260
  python3 tools/roofline_configs.py --collect $W/$cfg --config $cfg --out $O/${R}_roofline_$cfg.json
  cp $(find $W/$cfg/trace -name "*kernel_stats.csv" | head -1) $O/${R}_${cfg}_kernel_stats.csv
  echo "progress: $cfg done"
done
rm -rf $W
