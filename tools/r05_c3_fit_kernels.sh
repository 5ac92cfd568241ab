#!/bin/bash
# config 3 under the kernel trace: second moments by the pipelined register-block kernel and by the matrix-core kernel, E-step by
# the matrix cores and by the lane-per-row kernel
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call12; rm -rf $O; mkdir -p $O
for v in "0 0" "2 0"; do
  set -- $v
  export TEMPEST_AMD_COV_KERNEL=$1 TEMPEST_AMD_GMM_KERNEL=$2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$1_$2 -o s -- python3 tools/run_config.py c3 tpcn > $O/run_$1_$2.json 2> $O/run_$1_$2.err || { echo "run failed"; tail -5 $O/run_$1_$2.err; exit 1; }
  python3 - "$O/t_$1_$2" "$1 $2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_wcov", "k_gmm_estep", "k_multinomial", "k_mc_", "k_colsum2")):
        print(sys.argv[2], "|", n[:60].ljust(60), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:9.1f} us', f'{float(r["TotalDurationNs"])/1e6:8.1f} ms')
PY
  tail -1 $O/run_$1_$2.json | cut -c1-300
  find $O/t_$1_$2 -name "*kernel_trace.csv" -delete
done
