#!/bin/bash
# the bench line's digest (evidence, final ensemble, schedule) at N = 1 and -- all ranks on this one GPU over gloo, a REHEARSAL of the
# N > 1 control flow whose timings mean nothing -- at N = 2 and 4: the three digests must be equal
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05/digest; rm -rf $O; mkdir -p $O
FLAGS="--no-hip-callbacks --no-roofline --no-cpu-baseline --no-second-run --no-weak --steps 5 --warmup 2"
timeout -k 10 300 python3 bench.py $FLAGS > $O/n1.json 2> $O/n1.err || { echo "N=1 failed"; tail -5 $O/n1.err; exit 1; }
for N in 2 4; do
  TEMPEST_AMD_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N $FLAGS > $O/n$N.json 2> $O/n$N.err || { echo "N=$N failed"; tail -8 $O/n$N.err; exit 1; }
done
python3 - <<'PY'
import json
O="gpurun_out/r05/digest"
rows={}
for n in (1,2,4):
    line=[l for l in open(f"{O}/n{n}.json") if l.startswith("{")][-1]
    r=json.loads(line)
    rows[n]={"digest":r["digest"],"logz":r["logz"],"iterations_total":r["iterations_total"],"comm":{k:r.get("comm",{}).get(k) for k in ("world_size","backend","p2p_active_on_every_rank","p2p_by_rank")} if n>1 else None}
same=all(rows[n]["digest"]["ensemble_sha256"]==rows[1]["digest"]["ensemble_sha256"] and rows[n]["digest"]["logz_hex"]==rows[1]["digest"]["logz_hex"] and rows[n]["digest"]["schedule_sha256"]==rows[1]["digest"]["schedule_sha256"] for n in (2,4))
json.dump({"what":"bench.py digests at N = 1 (one process) and N = 2, 4 (TEMPEST_AMD_BENCH_REHEARSAL=1: all ranks on one GPU over gloo; 1 048 576 particles, seed 0)","equal":same,"runs":rows}, open(f"{O}/summary.json","w"), indent=1)
print("digests equal:", same)
for n in rows: print(n, rows[n]["digest"]["logz_hex"], rows[n]["digest"]["ensemble_sha256"][:16], rows[n]["digest"]["schedule_sha256"][:16])
PY
