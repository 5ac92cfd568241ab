"""The four step plugins of one Persistent Sampling iteration (reference: tempest/steps/), sharing one
StateManager: reweight -> train -> resample -> mutate (tempest/core.py:173-177)."""
from .reweight import Reweighter
from .train import Trainer
from .resample import Resampler
from .mutate import Mutator

__all__ = ["Reweighter", "Trainer", "Resampler", "Mutator"]
