"""The C-ABI library loads and exports every symbol include/mi355x_gan.h declares (no compute calls: no GPU here)."""
import os
import re

import gan_variant_research_amd as pkg
from gan_variant_research_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "mi355x_gan.h")).read()
    declared = set(re.findall(r"\b(gan_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.gan_version() >= 100


def test_errors_are_reported_not_swallowed():
    lib = _lib.load()
    rc = lib.gan_conv_igemm(None, None)
    assert rc != 0 and b"null" in lib.gan_last_error()


def test_missing_library_fails_loudly(tmp_path):
    import importlib
    import pytest
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.GanError):
            _lib.load(str(tmp_path / "nope.so"))
    finally:
        _lib._lib = saved
