#!/bin/bash
# FETCH_SIZE calibration; the tile-owner up-sampling count and the vectorised log-mixture update: tests, then the roofline table
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/call8; rm -rf $O; mkdir -p $O
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal -o c -- ./tools/ubench_fetch > $O/cal_expected.json 2> $O/cal.err
rc=$?; echo "calibration rc=$rc"
if [ $rc -ne 0 ]; then tail -5 $O/cal.err; exit $rc; fi
python3 tools/fetch_calibration.py $O/cal_expected.json $O/cal > $O/fetch_calibration.json; rm -rf $O/cal
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05/call8/fetch_calibration.json"))
for k,v in d["patterns"].items(): print(k, {a:b for a,b in v.items() if a.startswith("counter_over") or a=="duration_us"})
PY
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -x -k "multinomial or upsampl or history or logmix or mixture or fit_modes or fullsize or counts" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
mkdir -p $O/roof
TPH_ROOFLINE_META="$O/roof/meta.json" timeout -k 10 200 python3 tools/roofline_table.py > "$O/roof/plain.log" 2>&1 || { echo "roofline drive failed"; tail -5 $O/roof/plain.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/roof/trace" -o t -- python3 tools/roofline_table.py > "$O/roof/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
python3 - <<'PY'
import sys; sys.path.insert(0,"tools")
from roofline_table import _read
import numpy as np
dur,_=_read("gpurun_out/r05/call8/roof/trace")
for k,v in sorted(dur.items(), key=lambda kv:-np.mean(kv[1][-3:])):
    if k.startswith(("k_","void k_","void tph_scan","void rocprim")): print(f"{np.mean(v[-3:])/1e3:9.1f} us x{len(v):4d}  {k[:110]}")
PY
rm -rf $O/roof/trace
