"""INTEGRATION.md's ctypes stubs are executed as written: every ```python block of the document, in order,
in one namespace, on a small ensemble -- sections 1-6 on one GPU, section 7 (communicator attached, *_global entry points,
global resampling) on a one-rank gloo process group."""
import os
import re
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def test_document_has_the_sections():
    blocks = _blocks()
    assert len(blocks) == 7
    assert "tph_comm_attach" in blocks[6] and "tph_fit_modes_global" in blocks[6] and "tph_propose" in blocks[4]


def test_integration_stubs_run_as_written():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from tempest_amd import _lib
    _lib.load()                                                  # torch's HIP runtime first, then our library (see _lib.load)
    d, n = 4, 4096
    mean = torch.linspace(-1.0, 1.0, d, dtype=torch.float64, device="cuda:0")
    ns = {"LIB": _lib.LIB_PATH, "d": d, "n": n, "seed": 1234,
          "prior_transform": lambda u: 20.0 * u - 10.0,
          "log_likelihood": lambda x: -0.5 * ((x - mean) ** 2).sum(dim=1)}
    blocks = _blocks()
    for k, code in enumerate(blocks[:6]):
        exec(compile(code, f"INTEGRATION.md#section{k + 1}", "exec"), ns)
    assert ns["n_hist"] == 2 * n and 0 < ns["ess"] <= n * (1 + 1e-12)
    assert ns["samples"].shape == (ns["m_keep"], d) and abs(ns["weights"].sum() - 1.0) < 1e-10
    assert torch.isfinite(ns["means"]).all() and torch.isfinite(ns["cholinv"]).all()
    assert float(ns["state"][0]) == 3.0                          # three MCMC steps were adapted
    np.testing.assert_allclose(np.average(ns["samples"], weights=ns["weights"], axis=0), mean.cpu().numpy(), atol=0.5)
    # section 7 on a one-rank process group: the same library calls a multi-GPU launcher makes
    one_gpu = {k: ns[k].clone() for k in ("means", "covs")}
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        exec(compile(blocks[6], "INTEGRATION.md#section7", "exec"), ns)
    finally:
        dist.destroy_process_group()
    assert int(ns["claimed"].item()) == n                        # every global slot was claimed by exactly one rank
    assert ns["ok"].value == 1                                   # the peer-to-peer exchange attached and passed its self-test
    assert abs(ns["total"].value - ns["thr_host"][1]) < 1e-12    # global cumulative weight of the kept rows = kept_sum
    assert torch.isfinite(ns["means"]).all()
    # the global fit of a one-rank "cluster" is the one-GPU fit of the same multiplicities: same medians
    ctx, lib, P, I64, U64, U32, f64 = (ns[k] for k in ("ctx", "lib", "P", "I64", "U64", "U32", "f64"))
    m2, c2, l2, i2, w2 = f64(1, d), f64(1, d, d), f64(1, d, d), f64(1, d, d), f64(1, d, d)
    ns["chk"](lib.tph_fit_modes(ctx, P(ns["cnt"]), None, I64(ns["n_loc"]), 1, P(m2), P(c2), P(l2), P(i2), P(w2)))
    assert torch.equal(m2, ns["means"])
    np.testing.assert_allclose(ns["covs"].cpu().numpy(), c2.cpu().numpy(), rtol=1e-11, atol=1e-16)
    # the one-sided shuffle of a one-rank world is the plain gather of the selected rows
    ug, xg, lg = f64(d, n), f64(d, n), f64(n)
    ns["chk"](lib.tph_gather(ctx, P(ns["idx"]), I64(n), P(ug), P(xg), P(lg), I64(n)))
    assert torch.equal(ug, ns["u_new"]) and torch.equal(xg, ns["x_new"]) and torch.equal(lg, ns["l_new"])
    _ = one_gpu
    lib.tph_ctx_destroy(ctx)
