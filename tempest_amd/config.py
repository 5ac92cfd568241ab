"""Sampler configuration: same parameters, defaults, computed defaults and validation messages as the
reference's SamplerConfig (tempest/config.py:59-185; constants :232-242), plus keyword-only GPU
additions (device, backend, dtype is always FP64).  Error text is part of the drop-in contract: the
reference's tests assert on it (tests/test_config.py:69-229)."""
from __future__ import annotations

import warnings
from pathlib import Path
from typing import Any, Callable, List, Optional, Union

# Algorithm constants (tempest/config.py:232-242)
BETA_TOLERANCE: float = 1e-4
BETA_RTOL: float = 1e-8
ESS_TOLERANCE: float = 0.01
METRIC_ATOL: float = 0.5
METRIC_ATOL_CV: float = 0.01
DOF_FALLBACK: float = 1e6
TRIM_ESS: float = 0.99
TRIM_BINS: int = 1000

_FIELDS = (
    "prior_transform", "log_likelihood", "n_dim", "n_particles", "ess_ratio", "volume_variation",
    "log_likelihood_args", "log_likelihood_kwargs", "vectorize", "blobs_dtype", "periodic", "reflective",
    "pool", "clustering", "normalize", "cluster_every", "split_threshold", "n_max_clusters", "sample",
    "n_steps", "n_max_steps", "resample", "output_dir", "output_label", "random_state",
)
_GPU_FIELDS = ("device", "backend", "batch_prior", "graph", "student_em")


class SamplerConfig:
    """Immutable, validated configuration.  Attribute assignment after construction raises, like the
    reference's frozen dataclass."""

    def __init__(
        self,
        prior_transform: Callable,
        log_likelihood: Callable,
        n_dim: int,
        n_particles: Optional[int] = None,
        ess_ratio: float = 2.0,
        volume_variation: Optional[float] = None,
        log_likelihood_args: Optional[list] = None,
        log_likelihood_kwargs: Optional[dict] = None,
        vectorize: bool = False,
        blobs_dtype: Optional[str] = None,
        periodic: Optional[List[int]] = None,
        reflective: Optional[List[int]] = None,
        pool: Optional[Union[int, Any]] = None,
        clustering: bool = True,
        normalize: bool = True,
        cluster_every: int = 1,
        split_threshold: float = 1.0,
        n_max_clusters: Optional[int] = None,
        sample: str = "tpcn",
        n_steps: Optional[int] = None,
        n_max_steps: Optional[int] = None,
        resample: str = "mult",
        output_dir: Optional[Path] = None,
        output_label: Optional[str] = None,
        random_state: Optional[int] = None,
        *,
        device: Optional[Union[int, str]] = None,
        backend: str = "auto",
        batch_prior: Optional[bool] = None,
        graph: Optional[bool] = None,
        student_em: bool = False,
    ):
        put = lambda k, v: object.__setattr__(self, k, v)  # noqa: E731
        local = locals()
        for k in _FIELDS + _GPU_FIELDS:
            put(k, local[k])
        if not isinstance(n_dim, int):
            raise ValueError(f"n_dim must be int, got {type(n_dim).__name__}")
        # computed defaults (config.py:66-84)
        if output_dir is None:
            put("output_dir", Path("states"))
        elif isinstance(output_dir, str):
            put("output_dir", Path(output_dir))
        if output_label is None:
            put("output_label", "ps")
        if n_particles is None:
            put("n_particles", 2 * n_dim)
        if self.n_steps is None or self.n_steps <= 0:
            put("n_steps", 1)
        if self.n_max_steps is None or self.n_max_steps <= 0:
            put("n_max_steps", 20 * self.n_steps)
        self.validate()
        if self.volume_variation is not None and self.n_particles < self.n_dim + 1:
            warnings.warn(
                f"For dynamic mode, n_particles ({self.n_particles}) "
                f"should be >= n_dim + 1 ({self.n_dim + 1}) for reliable results. "
                f"Volume variation calculation may be inaccurate.",
                UserWarning,
                stacklevel=2,
            )
        put("_frozen", True)

    def __setattr__(self, key, value):
        if getattr(self, "_frozen", False) and key != "pool":
            raise AttributeError(f"cannot assign to field '{key}'")   # dataclasses.FrozenInstanceError is one
        object.__setattr__(self, key, value)

    # ------------------------------------------------------------------ validation
    def validate(self) -> None:
        bad = []
        if not callable(self.prior_transform):
            bad.append("prior_transform must be callable")
        if not callable(self.log_likelihood):
            bad.append("log_likelihood must be callable")
        if not isinstance(self.n_dim, int) or self.n_dim <= 0:
            bad.append(f"n_dim must be positive int, got {self.n_dim}")

        if not isinstance(self.n_particles, int):
            bad.append(f"n_particles must be int, got {type(self.n_particles)}")
        if isinstance(self.n_particles, (int, float)) and self.n_particles <= 0:
            bad.append(f"n_particles must be positive integer, got {self.n_particles}")

        if not isinstance(self.ess_ratio, (int, float)):
            bad.append(f"ess_ratio must be numeric, got {type(self.ess_ratio)}")
        elif self.ess_ratio <= 0:
            bad.append(f"ess_ratio must be positive, got {self.ess_ratio}")

        vv = self.volume_variation
        if vv is not None:
            if not isinstance(vv, (int, float)):
                bad.append(f"volume_variation must be numeric or None, got {type(vv)}")
            elif vv <= 0:
                bad.append(f"volume_variation ({vv}) must be positive")

        if self.sample not in ("tpcn", "rwm"):
            bad.append(f"Invalid sampler '{self.sample}': must be 'tpcn' or 'rwm'")
        if self.resample not in ("mult", "syst"):
            bad.append(f"Invalid resample '{self.resample}': must be 'mult' or 'syst'")
        if self.vectorize and self.blobs_dtype is not None:
            bad.append("Cannot vectorize likelihood with blobs")

        if self.periodic is not None and self.reflective is not None:
            both = set(self.periodic).intersection(set(self.reflective))
            if both:
                bad.append(f"Parameters cannot be both periodic and reflective: {both}")
        for name in ("periodic", "reflective"):
            idx = getattr(self, name)
            if idx is not None and not all(isinstance(i, int) and 0 <= i < self.n_dim for i in idx):
                bad.append(f"{name} indices must be integers in [0, {self.n_dim - 1}], got {idx}")

        if not isinstance(self.output_dir, Path):
            bad.append(f"output_dir must be Path, got {type(self.output_dir)}")
        if self.output_label is not None and not isinstance(self.output_label, str):
            bad.append(f"output_label must be str or None, got {type(self.output_label)}")
        if self.graph is not None and not isinstance(self.graph, bool):
            bad.append(f"graph must be bool or None, got {self.graph!r}")
        if not isinstance(self.student_em, bool):
            bad.append(f"student_em must be bool, got {self.student_em!r}")
        if self.backend not in ("auto", "torch", "numpy"):
            bad.append(f"backend must be 'auto', 'torch' or 'numpy', got {self.backend!r}")

        if bad:
            raise ValueError("Configuration validation failed:\n" + "\n".join(f"  - {m}" for m in bad))

    def get_target_metric(self) -> float:
        """ESS mode: ess_ratio * n_particles; dynamic mode: volume_variation (config.py:187-200)."""
        if self.volume_variation is not None:
            return self.volume_variation
        return self.ess_ratio * self.n_particles

    def to_dict(self) -> dict:
        out = {k: getattr(self, k) for k in _FIELDS}
        out["output_dir"] = str(self.output_dir)
        return out
