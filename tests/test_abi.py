"""CPU tests: the C-ABI library loads and exports every symbol include/nmx.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nmx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "nmx_paged_attention_v1" in syms and "nmx_gptq_marlin_gemm" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    from neuralmagic_vllm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libnmx_hip.so lacks: {missing}"
    lib.nmx_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.nmx_version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neuralmagic_vllm_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(ImportError):
        _lib.lib()


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under neuralmagic_vllm_amd/ may reference it."""
    pkg = os.path.join(ROOT, "neuralmagic_vllm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src, f"{f} references liboracle"
