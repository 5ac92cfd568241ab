#!/bin/bash
# round 5, GPU call 2: the canonical partition -- kernel parity tests, then the bitwise world-size tests
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_kernels_gpu.py tests/test_steps_gpu.py tests/test_sampler_gpu.py tests/test_hipcallbacks.py -q -x > $O/test_call2_kernels.log 2>&1
rc=$?; echo "kernel/steps/sampler tests rc=$rc"; tail -15 $O/test_call2_kernels.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 -m pytest tests/test_distributed.py -q -x -k "bitwise" -s > $O/test_call2_bitwise.log 2>&1
rc=$?; echo "bitwise tests rc=$rc"; tail -25 $O/test_call2_bitwise.log
exit $rc
