#!/bin/bash
# same box, alternating: the configurations end to end with round 4's package (scratch/r04_pkg: commit b07d541 + its library) and
# with this tree; two runs per process (the second is what a long job sees).  The old package is not kept in the tree; recreate it with
#   mkdir -p scratch/r04_pkg && git archive b07d541 tempest_amd tools/run_config.py oracle bench.py | tar -x -C scratch/r04_pkg
#   (cd scratch/r04_pkg/tempest_amd/csrc && make)      # or copy a library built from that commit to libtempest_hip.so there
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/ab_r04; rm -rf $O; mkdir -p $O
CFGS=${@:-"c2 tpcn" "c2 rwm" "c3 tpcn" "c5 tpcn"}
for rep in 1 2; do
  for cfg in "c2 tpcn" "c2 rwm" "c3 tpcn" "c5 tpcn"; do
    set -- $cfg
    (cd scratch/r04_pkg && TEMPEST_AMD_RUN_REPEAT=2 timeout -k 10 120 python3 tools/run_config.py $1 $2 2>> ../../$O/err_r04.log | grep "^{" | cut -c1-330 | sed "s/^/r04 /") >> $O/runs.txt || { echo "r04 run failed"; tail -3 $O/err_r04.log; exit 1; }
    (TEMPEST_AMD_RUN_REPEAT=2 timeout -k 10 120 python3 tools/run_config.py $1 $2 2>> $O/err_r05.log | grep "^{" | cut -c1-330 | sed "s/^/r05 /") >> $O/runs.txt || { echo "r05 run failed"; tail -3 $O/err_r05.log; exit 1; }
  done
done
python3 - <<'PY'
import json
for ln in open("gpurun_out/r05/ab_r04/runs.txt"):
    tag, js = ln.split(" ", 1)
    d = json.loads(js)
    print(tag, d["config"], d["kernel"], "run", d["run_in_process"], "wall", round(d["wall_s"], 3), "logz", d["logz"])
PY
