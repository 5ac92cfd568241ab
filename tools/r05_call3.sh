#!/bin/bash
# round 5, GPU call 3: the rest of the GPU suite after the canonical partition + the several-modes screened batches, then a bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_screened_gpu.py tests/test_api_gpu.py tests/test_cluster_gpu.py tests/test_state_manager_gpu.py tests/test_student_gpu.py tests/test_graph_gpu.py tests/test_fullsize_gpu.py tests/test_rccl_gpu.py tests/test_integration_doc.py tests/test_bench_contract.py tests/test_configs_gpu.py tests/test_distributed.py -m gpu -q -x --deselect tests/test_distributed.py::test_world_size_invariance_is_bitwise > $O/test_call3.log 2>&1
rc=$?; echo "rest of the suite rc=$rc"; tail -25 $O/test_call3.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --no-hip-callbacks > $O/bench_call3.json 2> $O/bench_call3.err
rc=$?; echo "bench rc=$rc"; head -c 700 $O/bench_call3.json
exit $rc
