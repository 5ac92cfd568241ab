#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 200 python3 tools/cold_warm.py c3 tpcn > $O/cold_warm_c3_b.json 2> $O/cold_warm_c3_b.err
echo "cold_warm c3 rc=$?"; cut -c1-1300 $O/cold_warm_c3_b.json
timeout -k 10 200 python3 tools/cold_warm.py c2 tpcn > $O/cold_warm_c2_b.json 2> $O/cold_warm_c2_b.err
echo "cold_warm c2 rc=$?"; cut -c1-900 $O/cold_warm_c2_b.json
timeout -k 10 500 python3 -m pytest tests/test_kernels_gpu.py tests/test_steps_gpu.py tests/test_sampler_gpu.py tests/test_api_gpu.py -q -x > $O/test_call6_a.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 $O/test_call6_a.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python3 -m pytest tests/test_distributed.py -q -x -k "bitwise" > $O/test_call6_b.log 2>&1
rc=$?; echo "bitwise tests rc=$rc"; tail -4 $O/test_call6_b.log | cut -c1-400
exit $rc
