#!/bin/bash
# streamers of train / resample after the rework: kernel tests, world invariance, then the traced roofline drive
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call9; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_state_manager_gpu.py tests/test_api_gpu.py tests/test_configs_gpu.py -q -x -m gpu -k "not 2097152" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert" $O/tests.log | head -20; exit $rc; fi
timeout -k 10 500 python3 -m pytest tests/test_distributed.py -q -x -m gpu -k "bitwise" > $O/tests_bitwise.log 2>&1
rc=$?; echo "bitwise rc=$rc"; tail -3 $O/tests_bitwise.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
mkdir -p $O/roof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/roof/trace" -o t -- python3 tools/roofline_table.py > "$O/roof/trace.log" 2>&1 || { echo "trace failed"; tail -5 $O/roof/trace.log; exit 1; }
python3 - <<'PY'
import sys; sys.path.insert(0,"tools")
from roofline_table import _read
import numpy as np
dur,_=_read("gpurun_out/r05/call9/roof/trace")
for k,v in sorted(dur.items(), key=lambda kv:-np.mean(kv[1][-3:])):
    if k.startswith(("k_","void k_","void tph_scan","void rocprim")) and np.mean(v[-3:])>15e3: print(f"{np.mean(v[-3:])/1e3:9.1f} us x{len(v):4d}  {k[:100]}")
PY
rm -rf $O/roof/trace
