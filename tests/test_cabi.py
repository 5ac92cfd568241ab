"""The C-ABI library loads and exports every symbol include/tempest_hip.h declares (no GPU needed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tempest_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tph_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_header():
    from tempest_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.tph_version() == 100


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    from tempest_amd import _lib
    with pytest.raises(_lib.TempestHipError):
        _lib.load(str(tmp_path / "nope.so"))
