#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
bash tools/r05_mf_ab.sh > $O/mf_ab.log 2>&1
rc=$?; echo "mf A/B rc=$rc"; tail -12 $O/mf_ab.log | cut -c1-1200
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 -m pytest tests/test_screened_gpu.py -q -x > $O/test_call4_screened.log 2>&1
rc=$?; echo "screened tests rc=$rc"; tail -4 $O/test_call4_screened.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench4 -o bench -- python3 bench.py --no-cpu-baseline --no-hip-callbacks --no-second-run > $O/bench_call4_prof.json 2> $O/bench_call4_prof.err
rc=$?; echo "profiled bench rc=$rc"
cp $(find $O/prof_bench4 -name "*kernel_stats.csv" | head -1) $O/bench_call4_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_bench4
head -c 300 $O/bench_call4_prof.json
exit $rc
