"""The JSON line bench.py prints (driver contract): checked on the line committed under profiles/ by the last GPU run, and on
the static parts of bench.py that do not need a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_unprofiled.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["timed_mcmc_steps"] * d["config"]["particles_global"] / (d["ms_per_step"] * d["steps"] * 1e-3)) \
        < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["logz"] - d["analytic_logz"]) < 0.5


def test_bench_cli_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout + out.stderr          # bench.py routes everything but the JSON line to stderr


import pytest


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    """The N > 1 control flow of bench.py (BASELINE config 4 sharded: strong scaling, max-over-ranks timing, the HIP-callback and
    weak-scaling legs, the peer-to-peer layer between two processes) walked on ONE GPU: TEMPEST_AMD_BENCH_REHEARSAL=1 puts every
    rank on cuda:0 over gloo.  The numbers mean nothing; the line must have the contract's shape and the sharded run must agree
    with the one-rank run of the same global ensemble on the evidence."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TEMPEST_AMD_BENCH_REHEARSAL="1", TEMPEST_AMD_P2P_TIMEOUT="60")
    common = ["--steps", "2", "--warmup", "1", "--particles", "32768", "--no-roofline", "--no-cpu-baseline"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    d2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and "REHEARSAL" in d2["data"]
    assert d2["config"]["particles_global"] == 32768 and d2["config"]["particles_per_gpu"] == 16384
    assert d2["hip_callbacks"]["value"] and d2["hip_callbacks"]["same_schedule_as_value_run"]
    assert d2["weak_scaling"]["particles_global"] == 65536 and d2["weak_scaling"]["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-hip-callbacks"] + common, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert d1["iterations_total"] == d2["iterations_total"] and d1["timed_mcmc_steps"] == d2["timed_mcmc_steps"]
    # the sharded sampler is the same sampler: equal to rounding at this size and seed (DESIGN.md section 7 describes the one
    # way two world sizes can part -- a rounding-level tie at the trim percentile -- and where it was seen)
    assert abs(d1["logz"] - d2["logz"]) < 1e-9
