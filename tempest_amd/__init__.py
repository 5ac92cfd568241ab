"""tempest_amd -- MI355X-native Persistent Sampling: the hot path of minaskar/tempest (reweight, resample,
MCMC mutation, proposal fit) as hand-written HIP kernels for gfx950 behind the reference's Sampler API."""
__version__ = "0.1.0"

from .sampler import Sampler

__all__ = ["Sampler", "HipCallbacks"]


def __getattr__(name):          # lazy: importing the package must not need hipcc or a GPU
    if name == "HipCallbacks":
        from .hipcallbacks import HipCallbacks
        return HipCallbacks
    raise AttributeError(name)
