"""SamplerConfig keeps the reference's defaults, computed defaults and validation messages
(tempest/config.py:59-185; the assertions follow the reference's tests/test_config.py).  CPU only."""
import warnings
from pathlib import Path

import pytest

from tempest_amd.config import (BETA_RTOL, BETA_TOLERANCE, DOF_FALLBACK, ESS_TOLERANCE, METRIC_ATOL, METRIC_ATOL_CV,
                                TRIM_BINS, TRIM_ESS, SamplerConfig)


def pt(u):
    return u


def ll(x):
    return 0.0


def test_defaults_and_computed_defaults():
    c = SamplerConfig(pt, ll, 2)
    assert (c.n_dim, c.n_particles, c.ess_ratio) == (2, 4, 2.0)          # n_particles = 2 * n_dim
    assert c.n_steps == 1 and c.n_max_steps == 20                          # code, not docs
    assert c.output_dir == Path("states") and c.output_label == "ps"
    assert c.resample == "mult" and c.sample == "tpcn" and c.clustering is True and c.vectorize is False
    c = SamplerConfig(pt, ll, 3, n_particles=100, n_steps=3, output_dir="custom_output", output_label="custom_label")
    assert c.n_particles == 100 and c.n_max_steps == 60
    assert c.output_dir == Path("custom_output") and c.output_label == "custom_label"
    assert (BETA_TOLERANCE, BETA_RTOL, ESS_TOLERANCE, METRIC_ATOL, METRIC_ATOL_CV, DOF_FALLBACK, TRIM_ESS, TRIM_BINS) == \
        (1e-4, 1e-8, 0.01, 0.5, 0.01, 1e6, 0.99, 1000)


@pytest.mark.parametrize("kw,msg", [
    (dict(sample="bogus"), "Invalid sampler 'bogus': must be 'tpcn' or 'rwm'"),
    (dict(resample="bogus"), "Invalid resample 'bogus': must be 'mult' or 'syst'"),
    (dict(vectorize=True, blobs_dtype="float"), "Cannot vectorize likelihood with blobs"),
    (dict(periodic=[0, 1], reflective=[1, 2]), "Parameters cannot be both periodic and reflective"),
    (dict(periodic=[0, 5]), "periodic indices must be integers in [0, 2]"),
    (dict(periodic=[-1]), "periodic indices must be integers in [0, 2]"),
    (dict(reflective=[3]), "reflective indices must be integers in [0, 2]"),
    (dict(ess_ratio=-1.0), "ess_ratio must be positive"),
    (dict(volume_variation=-0.5), "must be positive"),
    (dict(n_particles=0), "n_particles must be positive integer, got 0"),
])
def test_validation_messages(kw, msg):
    with pytest.raises(ValueError) as e:
        SamplerConfig(pt, ll, 3, **kw)
    assert msg in str(e.value)
    assert str(e.value).startswith("Configuration validation failed:\n  - ")


def test_n_dim_type_and_sign():
    with pytest.raises(ValueError, match="n_dim must be int"):
        SamplerConfig(pt, ll, 2.5)
    with pytest.raises(ValueError, match="n_dim must be positive int"):
        SamplerConfig(pt, ll, 0)
    with pytest.raises(ValueError, match="prior_transform must be callable"):
        SamplerConfig(None, ll, 2)


def test_errors_are_aggregated():
    with pytest.raises(ValueError) as e:
        SamplerConfig(pt, ll, 3, sample="x", resample="y", ess_ratio=0)
    assert str(e.value).count("\n  - ") == 3


def test_frozen_and_target_metric():
    c = SamplerConfig(pt, ll, 2, n_particles=50)
    with pytest.raises(AttributeError):
        c.n_particles = 10
    assert c.n_particles == 50 and c.get_target_metric() == 100.0
    assert SamplerConfig(pt, ll, 2, n_particles=50, volume_variation=0.25).get_target_metric() == 0.25
    assert c.to_dict()["output_dir"] == "states"


def test_dynamic_mode_warning():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        SamplerConfig(pt, ll, 10, n_particles=5, volume_variation=0.1)
    assert any("dynamic mode" in str(x.message) and "n_particles" in str(x.message) for x in w)


def test_package_exports():
    """reference tests/test_package_install.py:20-28."""
    import tempest_amd
    from tempest_amd import Sampler
    from tempest_amd.steps import Mutator, Resampler, Reweighter, Trainer
    assert tempest_amd.Sampler is Sampler and all(callable(c) for c in (Mutator, Resampler, Reweighter, Trainer))
