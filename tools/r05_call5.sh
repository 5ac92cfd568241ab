#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_steps_gpu.py tests/test_sampler_gpu.py -q -x > $O/test_call5_a.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 $O/test_call5_a.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python3 -m pytest tests/test_distributed.py -q -x -k "bitwise and (2-True or 3-True)" > $O/test_call5_b.log 2>&1
rc=$?; echo "bitwise tests rc=$rc"; tail -4 $O/test_call5_b.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --no-hip-callbacks > $O/bench_call5.json 2> $O/bench_call5.err
rc=$?; echo "bench rc=$rc"; head -c 400 $O/bench_call5.json; echo
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python3 tools/cold_warm.py c2 tpcn > $O/cold_warm_c2.json 2> $O/cold_warm_c2.err
echo "cold_warm c2 rc=$?"; cut -c1-1500 $O/cold_warm_c2.json
timeout -k 10 200 python3 tools/cold_warm.py c3 tpcn > $O/cold_warm_c3.json 2> $O/cold_warm_c3.err
echo "cold_warm c3 rc=$?"; cut -c1-1500 $O/cold_warm_c3.json
