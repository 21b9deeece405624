"""Quantization method registry — mirror of vllm/model_executor/layers/quantization/__init__.py."""
from typing import Dict, Type

from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
from neuralmagic_vllm_amd.layers.quantization.base_config import QuantizationConfig
from neuralmagic_vllm_amd.layers.quantization.compressed_tensors import CompressedTensorsConfig
from neuralmagic_vllm_amd.layers.quantization.fp8 import Fp8Config
from neuralmagic_vllm_amd.layers.quantization.gptq import GPTQConfig
from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig
from neuralmagic_vllm_amd.layers.quantization.gptq_marlin_24 import GPTQMarlin24Config
from neuralmagic_vllm_amd.layers.quantization.marlin import MarlinConfig

QUANTIZATION_METHODS: Dict[str, Type[QuantizationConfig]] = {
    "awq": AWQConfig,
    "compressed-tensors": CompressedTensorsConfig,
    "fp8": Fp8Config,
    "gptq_marlin_24": GPTQMarlin24Config,  # before gptq_marlin / gptq: override order (__init__.py:32-36)
    "gptq": GPTQConfig,
    "gptq_marlin": GPTQMarlinConfig,  # must stay before plain gptq in override order (reference comment, __init__.py:32-36)
    "marlin": MarlinConfig,
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return QUANTIZATION_METHODS[quantization]


__all__ = ["QuantizationConfig", "get_quantization_config", "QUANTIZATION_METHODS"]
