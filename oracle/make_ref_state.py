"""Generate the reference-WRITTEN checkpoint fixtures by importing THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Usage (scratch cwd; the reference is read-only and never travels):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python3 /root/repo/oracle/make_ref_state.py

Runs the reference sampler (minaskar/tempest v0.2.1) for a few iterations on a small Gaussian target and lets the
REFERENCE write its own files:
  tests/golden/ref_state_small.state        StateManager.save_state (state_manager.py:597-633): dill of to_dict()
  tests/golden/ref_state_small_core.state   the dict of SamplerCore.save_sampler_state (core.py:249-279) WITHOUT its
                                            "sampler" entry (dill.dumps of the core object: pickled callables and classes
                                            of the reference package -- code, not data, and not loadable without it)
  tests/golden/ref_state_small.npz          what the reference computes from that state: flat history arrays,
                                            compute_logw_and_logz(1.0), the iteration table
  tests/golden/ref_state_small_posterior.npz   (--posterior-only) what the reference's posterior() returns on that state
                                            (core.py:187-242): trimmed and untrimmed, with logw, and a second trim setting
Both .state files contain only dicts, lists, floats and NumPy arrays.
"""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main(posterior_only=False):
    import dill
    import tempest as tp
    d, n = 3, 64
    mean = np.array([-1.0, 0.5, 2.0])

    def prior(u):
        return 20.0 * u - 10.0

    def loglike(x):
        return -0.5 * np.sum((x - mean) ** 2, axis=-1) - 0.5 * d * np.log(2 * np.pi)
    np.random.seed(123)
    s = tp.Sampler(prior, loglike, d, n_particles=n, vectorize=True, clustering=False, random_state=123)
    core = s._core if hasattr(s, "_core") else s.core
    core._initialize_fresh()                      # what run() does before its loop (core.py:128)
    from tempest.tools import ProgressBar
    core.pbar = ProgressBar(False)
    core.reweighter.pbar = core.pbar
    core.trainer.pbar = core.pbar
    core.mutator.pbar = core.pbar
    for _ in range(7):
        s.sample()
    st = s.state
    if posterior_only:
        # the same seeded run again: it must be the state the committed fixtures hold, then the reference's own posterior()
        old = np.load(os.path.join(OUT, "ref_state_small.npz"))
        assert np.array_equal(old["u"], st.get_history("u", flat=True)) and np.array_equal(old["logl"], st.get_history("logl", flat=True))
        out = {}
        for tag, kw in (("trim", dict(trim_importance_weights=True)), ("full", dict(trim_importance_weights=False)),
                        ("trim90", dict(trim_importance_weights=True, ess_trim=0.9, bins_trim=50))):
            x, w, logl, logw = s.posterior(return_logw=True, **kw)
            out.update({f"x_{tag}": x, f"w_{tag}": w, f"logl_{tag}": logl, f"logw_{tag}": logw})
        np.savez(os.path.join(OUT, "ref_state_small_posterior.npz"), **out)
        print("wrote ref_state_small_posterior.npz", {k: v.shape for k, v in out.items()})
        return
    path = os.path.join(OUT, "ref_state_small.state")
    st.save_state(path)
    dd = st.to_dict()
    dd["random_state"] = core.config.random_state
    dd["n_total"] = getattr(core, "n_total", None)
    dd["logz_err"] = getattr(core, "logz_err", None)
    with open(os.path.join(OUT, "ref_state_small_core.state"), "wb") as f:      # core.py:249-279 minus d["sampler"]
        dill.dump(dd, f)
    logw, logz = st.compute_logw_and_logz(1.0)
    logw_half, logz_half = st.compute_logw_and_logz(0.5)
    np.savez(os.path.join(OUT, "ref_state_small.npz"),
             u=st.get_history("u", flat=True), x=st.get_history("x", flat=True), logl=st.get_history("logl", flat=True),
             beta=np.array(st.get_history("beta")), logz_t=np.array(st.get_history("logz")),
             iter=np.array(st.get_history("iter")), calls=np.array(st.get_history("calls")),
             steps=np.array(st.get_history("steps")), ess=np.array(st.get_history("ess")),
             cur_u=st.get_current("u"), cur_logl=st.get_current("logl"), cur_beta=st.get_current("beta"),
             cur_iter=st.get_current("iter"), cur_calls=st.get_current("calls"),
             logw1=logw, logz1=logz, logw_half=logw_half, logz_half=logz_half, n_particles=n, n_dim=d, mean=mean)
    print("wrote", path, "iterations", len(st.get_history("beta")), "beta", st.get_current("beta"), "logz1", logz)
    # the files must not need the reference to be read back
    with open(path, "rb") as f:
        back = dill.load(f)
    assert set(back) == {"_current", "_history", "n_dim"}, set(back)


if __name__ == "__main__":
    import sys
    main(posterior_only="--posterior-only" in sys.argv)
