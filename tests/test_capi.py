"""The C-ABI library loads and exports every symbol include/ofx.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from detprocess_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ofx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofx_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SYMBOL_NAMES)


def test_library_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with: make -C detprocess_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().ofx_last_error() is not None


def test_product_path_has_no_cpu_fallback(monkeypatch, tmp_path):
    """If the extension is missing the product raises instead of computing on the CPU."""
    import pytest
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OfxError):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "detprocess_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f
