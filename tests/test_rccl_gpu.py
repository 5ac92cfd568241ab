"""The sharded code path over the REAL RCCL backend (torch.distributed "nccl") at world size 1: every collective the
multi-GPU run issues (all-reduce of the step statistics, all-gather of the reweight triples, the all-to-all-v of the
resample shuffle, the sharded checkpoint barrier) goes through RCCL, and the run must reproduce the un-sharded one bit for
bit (same counter-based draws, one shard = the whole ensemble)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
import tempest_amd as tp

def prior(u): return 20 * u - 10
def like(x): return -(10.0 * (x[:, ::2] ** 2 - x[:, 1::2]) ** 2 + (x[:, ::2] - 1.0) ** 2).sum(dim=1)

out = {}
for sharded in (False, True):
    if sharded:
        os.environ["TEMPEST_AMD_FORCE_COMM"] = "1"
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", world_size=1, rank=0,
                                device_id=torch.device("cuda", 0))
    for kernel, resample in (("tpcn", "mult"), ("rwm", "syst")):
        s = tp.Sampler(prior, like, 4, n_particles=2048, vectorize=True, clustering=False, random_state=3,
                       sample=kernel, resample=resample, device=0)
        assert (s.state.comm is not None and s.state.comm.active) == sharded
        s.run(n_total=8192, progress=False)
        x, w, _ = s.posterior()
        out["%%s_%%s_%%d" %% (kernel, resample, sharded)] = [s.evidence()[0], int(len(s.state.get_history("beta"))),
                                                          float(np.average(x[:, 0], weights=w)), int(len(w))]
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_sharded_path_over_rccl_world1_equals_unsharded():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.pop("TEMPEST_AMD_FORCE_COMM", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT, "port": port}], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    for key in ("tpcn_mult", "rwm_syst"):
        a, b = res[key + "_0"], res[key + "_1"]
        # evidence, iteration count and posterior size to the last bit; the weighted mean to rounding (the posterior weights are
        # normalised by the kept sum of the trim, which the one-GPU function forms from segment sums of the sorted weights and
        # the global one from the counters of its radix select: two summation orders)
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (key, a, b)
        assert abs(a[2] - b[2]) <= 1e-12 * abs(a[2]), (key, a, b)
        truth = 2 * (np.log(np.pi / np.sqrt(10.0)) - np.log(400.0))
        assert abs(a[0] - truth) < 0.6      # the algorithm's own positive bias is ~0.2-0.3 here
