#!/bin/bash
# full gpu suite with the staged blocked kernel on by default, then config 5 at full size (whole-run effect of the staging)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call7; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
rc=$?; echo "gpu suite rc=$rc"; tail -5 $O/gpu_suite.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python3 tools/run_c5_full.py > $O/c5_full.json 2> $O/c5_full.err
rc=$?; echo "c5 rc=$rc"; cut -c1-700 $O/c5_full.json
