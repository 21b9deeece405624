"""Host-side mirror of the reference's quantization plugin interface
(vllm/model_executor/layers/quantization/base_config.py:8-97, vllm/model_executor/layers/linear.py:69-100,
vllm/model_executor/utils.py set_weight_attrs). Same names, argument meaning and error behaviour, so a
`QuantizeMethodBase` written against the reference runs unchanged on top of `neuralmagic_vllm_amd._custom_ops`."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch


def set_weight_attrs(weight: torch.Tensor, weight_attrs: Optional[Dict[str, Any]]) -> None:
    """Attach loader metadata (input_dim / output_dim / packed_dim / pack_factor / weight_loader ...) to a parameter.
    The reference's TP weight loaders narrow checkpoint tensors with exactly these attributes (linear.py:383-470)."""
    if weight_attrs is None:
        return
    for key, value in weight_attrs.items():
        assert not hasattr(weight, key), f"Overwriting existing tensor attribute: {key}"
        setattr(weight, key, value)


class QuantizeMethodBase(ABC):
    """Base class for different quantized methods."""

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return


class LinearMethodBase(QuantizeMethodBase):
    """create_weights(layer, input_size_per_partition, output_partition_sizes, input_size, output_size, params_dtype,
    **extra_weight_attrs); apply(layer, x, bias) — linear.py:69-100."""

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class QuantizationConfig(ABC):

    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @classmethod
    def get_min_capability(cls) -> int:
        # gfx950 reports (9, 5) -> 95 (vllm/platforms/rocm.py); every method in this package runs on it
        return 0

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError

    def get_scaled_act_names(self) -> List[str]:
        return []


def replace_tensor(layer: torch.nn.Module, name: str, new_t: torch.Tensor) -> None:
    """Swap a registered parameter's storage in place (the reference uses resize_ + copy_ so that the buffer the
    engine already tracks is reused: gptq_marlin.py:389-397)."""
    getattr(layer, name).resize_(new_t.shape)
    getattr(layer, name).copy_(new_t)
