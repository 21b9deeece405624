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


def test_compressed_tensors_scheme_dispatch():
    """Host logic only (no GPU): the compressed-tensors config picks the scheme the reference picks
    (compressed_tensors.py:132-160) and matches targets by class name or regex (utils.py:78-122)."""
    import torch
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors import (
        CompressedTensorsConfig, CompressedTensorsW4A16Sparse24, CompressedTensorsWNA16, find_first_name_or_class_match)
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors_w8a8 import CompressedTensorsW8A8

    class Linear(torch.nn.Module):
        pass

    def cfg(fmt, w, a=None):
        return CompressedTensorsConfig.from_config({"format": fmt, "config_groups": {
            "g": {"targets": ["Linear"], "weights": w, "input_activations": a}}})

    w4 = {"num_bits": 4, "symmetric": True, "strategy": "group", "group_size": 128}
    assert isinstance(cfg("pack-quantized", w4).get_scheme(Linear()), CompressedTensorsWNA16)
    assert isinstance(cfg("marlin-24", w4).get_scheme(Linear()), CompressedTensorsW4A16Sparse24)
    w8 = {"num_bits": 8, "symmetric": True, "strategy": "tensor"}
    st = cfg("int-quantized", w8, {"num_bits": 8, "symmetric": True, "strategy": "tensor"}).get_scheme(Linear())
    assert isinstance(st, CompressedTensorsW8A8) and st.is_static_input_scheme
    with pytest.raises(NotImplementedError):
        cfg("marlin-24", {"num_bits": 8, "symmetric": True, "strategy": "channel"}).get_scheme(Linear())
    with pytest.raises(ValueError):
        CompressedTensorsWNA16("group", 4, None)
    assert find_first_name_or_class_match("model.layers.0.q_proj", Linear(), ["re:.*q_proj"]) == "re:.*q_proj"
    assert find_first_name_or_class_match("x", Linear(), ["linear"], check_contains=True) == "linear"
    assert find_first_name_or_class_match("x", Linear(), ["Conv"]) is None
