#!/bin/bash
# matrix-core E-step and the pipelined covariance tile: tests; then which kernel for the second moments and where the sorted
# up-sampling count starts to pay
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call11; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_cluster_gpu.py tests/test_student_gpu.py -q -x -m gpu > $O/tests_cluster.log 2>&1
rc=$?; echo "cluster tests rc=$rc"; tail -5 $O/tests_cluster.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E " $O/tests_cluster.log | head -20; exit $rc; fi
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "fit or cov or moment" > $O/tests_fit.log 2>&1
rc=$?; echo "fit tests rc=$rc"; tail -3 $O/tests_fit.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E " $O/tests_fit.log | head -20; exit $rc; fi
timeout -k 10 300 python3 tools/bench_upsample_counts.py > $O/upsample_ab.jsonl 2> $O/upsample_ab.err
rc=$?; echo "upsample A/B rc=$rc"; cat $O/upsample_ab.jsonl
