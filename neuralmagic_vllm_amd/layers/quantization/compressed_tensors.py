"""compressed-tensors checkpoints — mirror of vllm/model_executor/layers/quantization/compressed_tensors/
(compressed_tensors.py:19-234 config + linear method, utils.py:9-122 argument model / target matching,
schemes/compressed_tensors_{wNa16,w4a16_24,w8a8,unquantized}.py). The scheme picked per layer decides which op of
the hot path runs: pack-quantized int4/int8 -> gptq_marlin_repack + gptq_marlin_gemm, marlin-24 -> gptq_marlin_24_gemm,
int-quantized W8A8 -> scaled_int8_quant + cutlass_scaled_mm."""
import re
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.nn.functional as F
from torch.nn import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import LinearMethodBase, QuantizationConfig, set_weight_attrs
from neuralmagic_vllm_amd.layers.quantization.compressed_tensors_w8a8 import CompressedTensorsW8A8
from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import (GPTQ_MARLIN_MAX_PARALLEL, GPTQ_MARLIN_MIN_THREAD_N,
                                                                  GPTQMarlinState, marlin_permute_scales)
from neuralmagic_vllm_amd.layers.quantization.gptq_marlin_24 import (GPTQ_MARLIN_24_MAX_PARALLEL,
                                                                     GPTQ_MARLIN_24_MIN_THREAD_N)

WNA16_SUPPORTED_BITS = [4, 8]
W4A16SPARSE24_SUPPORTED_BITS = [4]


@dataclass
class QuantizationArgs:  # utils.py:38-75 (pydantic model in the reference)
    num_bits: int = 8
    type: str = "int"
    symmetric: bool = True
    group_size: Optional[int] = None
    strategy: Optional[str] = None
    block_structure: Optional[str] = None
    dynamic: bool = False
    observer: str = "minmax"
    observer_kwargs: Optional[Dict[str, Any]] = None

    @classmethod
    def parse_obj(cls, obj: Optional[Dict[str, Any]]) -> "QuantizationArgs":
        if obj is None:
            raise ValueError("no quantization arguments")
        known = {k: v for k, v in obj.items() if k in cls.__dataclass_fields__}
        return cls(**known)


def find_first_name_or_class_match(name: str, module: torch.nn.Module, targets, check_contains: bool = False):
    """utils.py:78-122: first target matching the layer name, else the layer's class name ("re:" = regex)."""

    def first(value: str):
        for target in targets:
            if target.startswith("re:"):
                if re.match(target[3:], value):
                    return target
            elif check_contains:
                if target.lower() in value.lower():
                    return target
            elif target == value:
                return target
        return None

    return first(name) or first(module.__class__.__name__)


class CompressedTensorsUnquantized:  # schemes/compressed_tensors_unquantized.py

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        pass

    def create_weights(self, layer, output_partition_sizes, input_size_per_partition, params_dtype, weight_loader=None,
                       **kwargs):
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "weight_loader": weight_loader})
        layer.register_parameter("weight", weight)

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor):
        return F.linear(x, layer.weight)


class CompressedTensorsWNA16:  # schemes/compressed_tensors_wNa16.py:18-175

    def __init__(self, strategy: str, num_bits: int, group_size: Optional[int] = None):
        self.num_bits, self.strategy, self.group_size = num_bits, strategy, group_size
        if self.strategy == "group" and self.group_size is None:
            raise ValueError("group_size must be given when using strategy group")

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        pass

    def create_weights(self, layer, input_size: int, output_partition_sizes: List[int], input_size_per_partition: int,
                       params_dtype: torch.dtype, weight_loader: Optional[Callable] = None, **kwargs):
        pack_factor = 32 // self.num_bits
        out_pp = sum(output_partition_sizes)
        group_size = self.group_size if self.group_size is not None else input_size
        weight_scale_dim, n_groups = None, input_size // group_size
        if input_size != input_size_per_partition and self.group_size is not None:
            weight_scale_dim, n_groups = 1, input_size_per_partition // group_size
        dev = kwargs.get("device", None)
        weight = Parameter(torch.empty(out_pp, input_size_per_partition // pack_factor, dtype=torch.int32, device=dev),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "packed_dim": 1, "pack_factor": pack_factor,
                                  "weight_loader": weight_loader})
        layer.register_parameter("weight_packed", weight)
        weight_scale = Parameter(torch.empty(out_pp, n_groups, dtype=params_dtype, device=dev), requires_grad=False)
        set_weight_attrs(weight_scale, {"weight_loader": weight_loader, "input_dim": weight_scale_dim, "output_dim": 0})
        layer.register_parameter("weight_scale", weight_scale)
        weight_shape = Parameter(torch.empty(2, dtype=torch.int64, device=dev), requires_grad=False)
        layer.register_parameter("weight_shape", weight_shape)
        set_weight_attrs(weight_shape, {"weight_loader": weight_loader, "ignore_warning": True})
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = out_pp
        layer.input_size = input_size
        layer.marlin_state = GPTQMarlinState.REPACK
        layer.is_k_full = True
        layer.group_size = group_size
        layer.workspace = torch.zeros((out_pp // GPTQ_MARLIN_MIN_THREAD_N) * GPTQ_MARLIN_MAX_PARALLEL, dtype=torch.int,
                                      device=dev, requires_grad=False)

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor):
        reshaped_x = x.reshape(-1, x.shape[-1])
        size_m = reshaped_x.shape[0]
        part_n, part_k = layer.output_size_per_partition, layer.input_size_per_partition
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            dev = layer.weight_packed.device
            layer.g_idx = Parameter(torch.empty(0, dtype=torch.int, device=dev), requires_grad=False)
            layer.g_idx_sort_indices = Parameter(torch.empty(0, dtype=torch.int, device=dev), requires_grad=False)
            # checkpoint layout [N, K/pack] -> GPTQ layout [K/pack, N] -> Marlin
            marlin_qweight = ops.gptq_marlin_repack(layer.weight_packed.t().contiguous(), layer.g_idx_sort_indices, part_k,
                                                    part_n, self.num_bits)
            layer.weight_packed = Parameter(marlin_qweight, requires_grad=False)
            scales = layer.weight_scale.reshape(part_n, -1).t().contiguous()
            layer.weight_scale = Parameter(marlin_permute_scales(scales, part_k, part_n, layer.group_size, self.num_bits),
                                           requires_grad=False)
            layer.workspace = layer.workspace.to(dev)
        out = ops.gptq_marlin_gemm(reshaped_x, layer.weight_packed, layer.weight_scale, layer.g_idx,
                                   layer.g_idx_sort_indices, layer.workspace, self.num_bits, size_m, part_n, part_k,
                                   layer.is_k_full)
        return out.reshape(x.shape[:-1] + (part_n, ))


class CompressedTensorsW4A16Sparse24:  # schemes/compressed_tensors_w4a16_24.py:15-138

    def __init__(self, strategy: str, num_bits: int, group_size: Optional[int] = None):
        self.strategy, self.group_size, self.num_bits, self.tile_size = strategy, group_size, num_bits, 16
        if self.strategy == "group" and self.group_size is None:
            raise ValueError("group_size must be given when using strategy group")

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        pass

    def create_weights(self, layer, input_size: int, output_partition_sizes: List[int], input_size_per_partition: int,
                       params_dtype: torch.dtype, weight_loader: Optional[Callable] = None, **kwargs):
        pack_factor = 32 // self.num_bits
        out_pp = sum(output_partition_sizes)
        dev = kwargs.get("device", None)
        qweight = Parameter(torch.empty(input_size_per_partition // self.tile_size // 2,
                                        out_pp * self.tile_size // pack_factor, dtype=torch.int32, device=dev),
                            requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": pack_factor,
                                   "marlin_tile_size": self.tile_size, "weight_loader": weight_loader})
        layer.register_parameter("weight_packed", qweight)
        input_groups = 1 if self.group_size is None else input_size_per_partition // self.group_size
        scales = Parameter(torch.empty(input_groups, out_pp, dtype=params_dtype, device=dev), requires_grad=False)
        set_weight_attrs(scales, {"output_dim": 1, "input_dim": None if input_groups == 1 else 0,
                                  "weight_loader": weight_loader})
        layer.register_parameter("scale_packed", scales)
        weight_shape = Parameter(torch.empty(2, dtype=torch.int64, device=dev), requires_grad=False)
        layer.register_parameter("weight_shape", weight_shape)
        set_weight_attrs(weight_shape, {"weight_loader": weight_loader})
        meta = Parameter(torch.empty(input_size_per_partition // 8 // 2 // 2, out_pp * 2, dtype=torch.int16, device=dev),
                         requires_grad=False)
        set_weight_attrs(meta, {"input_dim": 0, "packed_dim": 1, "pack_factor": 1, "output_dim": 1, "marlin_tile_size": 2,
                                "weight_loader": weight_loader})
        layer.register_parameter("meta", meta)
        layer.workspace = Parameter(torch.zeros((out_pp // GPTQ_MARLIN_24_MIN_THREAD_N) * GPTQ_MARLIN_24_MAX_PARALLEL,
                                                dtype=torch.int, device=dev), requires_grad=False)

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor):
        x_2d = x.view(-1, x.shape[-1])
        out = ops.gptq_marlin_24_gemm(x_2d, layer.weight_packed, layer.meta, layer.scale_packed, layer.workspace,
                                      self.num_bits, x_2d.shape[0], layer.scale_packed.shape[1], x_2d.shape[1])
        return out.view(x.shape[:-1] + (out.shape[1], ))


class CompressedTensorsConfig(QuantizationConfig):

    def __init__(self, layer_quant_details: Dict[str, Any], ignore: List[str], quant_format: str):
        self.ignore = ignore
        self.layer_quant_details = layer_quant_details
        self.quant_format = quant_format

    @classmethod
    def get_name(cls) -> str:
        return "compressed_tensors"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CompressedTensorsConfig":
        details: Dict[str, Any] = {}
        for _, group in config["config_groups"].items():
            for target in group.get("targets"):
                details[target] = {"weights": QuantizationArgs.parse_obj(group.get("weights"))}
                try:
                    details[target]["input_activations"] = QuantizationArgs.parse_obj(group.get("input_activations"))
                except Exception:
                    details[target]["input_activations"] = None
        return cls(layer_quant_details=details, ignore=config.get("ignore", None), quant_format=config.get("format", None))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["CompressedTensorsLinearMethod"]:
        return CompressedTensorsLinearMethod(self)

    # compressed_tensors.py:94-160
    @staticmethod
    def _is_static_tensor_w8a8(w: QuantizationArgs, a: QuantizationArgs) -> bool:
        return (w.num_bits == a.num_bits == 8 and w.strategy in ("tensor", "channel") and a.strategy == "tensor"
                and w.symmetric and a.symmetric and not w.dynamic and not a.dynamic)

    @staticmethod
    def _is_dynamic_token_w8a8(w: QuantizationArgs, a: QuantizationArgs) -> bool:
        return (w.num_bits == a.num_bits == 8 and w.strategy in ("tensor", "channel") and a.strategy == "token"
                and w.symmetric and a.symmetric and not w.dynamic and a.dynamic)

    @staticmethod
    def _is_wNa16_group_channel(w: QuantizationArgs, a: Optional[QuantizationArgs]) -> bool:
        return a is None and w.strategy in ("channel", "group") and w.symmetric and not w.dynamic

    def _get_schema(self, weight_quant: QuantizationArgs, input_quant: Optional[QuantizationArgs]):
        if self._is_wNa16_group_channel(weight_quant, input_quant):
            if self.quant_format == "marlin-24" and weight_quant.num_bits in W4A16SPARSE24_SUPPORTED_BITS:
                return CompressedTensorsW4A16Sparse24(strategy=weight_quant.strategy, num_bits=weight_quant.num_bits,
                                                      group_size=weight_quant.group_size)
            if self.quant_format == "pack-quantized" and weight_quant.num_bits in WNA16_SUPPORTED_BITS:
                return CompressedTensorsWNA16(num_bits=weight_quant.num_bits, strategy=weight_quant.strategy,
                                              group_size=weight_quant.group_size)
        if self.quant_format == "int-quantized" and input_quant is not None:
            if self._is_static_tensor_w8a8(weight_quant, input_quant):
                return CompressedTensorsW8A8(strategy=weight_quant.strategy, is_static_input_scheme=True)
            if self._is_dynamic_token_w8a8(weight_quant, input_quant):
                return CompressedTensorsW8A8(strategy=weight_quant.strategy, is_static_input_scheme=False)
        raise NotImplementedError("No compressed-tensors compatible scheme was found.")

    def get_scheme(self, layer: torch.nn.Module, name: str = ""):
        target = find_first_name_or_class_match(name=name, module=layer, targets=self.layer_quant_details.keys(),
                                                check_contains=True)
        if target is None:
            raise ValueError(f"Could not matching target for layer {layer}")
        d = self.layer_quant_details[target]
        return self._get_schema(weight_quant=d["weights"], input_quant=d["input_activations"])


class CompressedTensorsLinearMethod(LinearMethodBase):

    def __init__(self, quantization_config: CompressedTensorsConfig):
        self.quantization_config = quantization_config

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return layer.scheme.process_weights_after_loading(layer)

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        scheme = self.quantization_config.get_scheme(layer=layer)
        scheme.create_weights(layer=layer, input_size=input_size, input_size_per_partition=input_size_per_partition,
                              output_partition_sizes=output_partition_sizes, output_size=output_size,
                              params_dtype=params_dtype, weight_loader=extra_weight_attrs.get("weight_loader"),
                              device=extra_weight_attrs.get("device"))
        layer.scheme = scheme

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None):
        if bias is not None:
            raise ValueError("bias is not supported for this linear method")
        if getattr(layer, "scheme", None) is None:
            raise ValueError("A scheme must be defined for each layer")
        return layer.scheme.apply_weights(layer, x)
