#!/bin/bash
# after the small-configuration fixes (K2 floor, fold per beta, wide column sums, covariance fill above 32-D): tests, then the
# same-box A/B against round 4's package
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call13; rm -rf $O; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests/test_kernels_gpu.py tests/test_steps_gpu.py tests/test_sampler_gpu.py tests/test_cluster_gpu.py -q -x -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "^E " $O/tests.log | head -20; exit $rc; fi
bash tools/r05_ab_r04.sh
