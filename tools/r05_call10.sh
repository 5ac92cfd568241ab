#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call10; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_graph_gpu.py -q -x -m gpu -k "probe_driven" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -30 $O/tests.log | cut -c1-400
