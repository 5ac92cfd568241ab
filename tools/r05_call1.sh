#!/bin/bash
# round 5, GPU call 1: MFMA summation-order probe, config 5 at full size on one GPU, the mapped-history tests, a bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_history_vm_gpu.py -x -q > $O/test_history_vm.log 2>&1
rc=$?; echo "history tests rc=$rc"; tail -5 $O/test_history_vm.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 tools/run_c5_full.py > $O/c5_full.json 2> $O/c5_full.err
rc=$?; echo "c5 full rc=$rc"; tail -3 $O/c5_full.err; head -c 1500 $O/c5_full.json
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python3 bench.py > $O/bench_call1.json 2> $O/bench_call1.err
rc=$?; echo "bench rc=$rc"; head -c 1200 $O/bench_call1.json
exit $rc
