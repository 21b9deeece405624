"""GPTQ (exllama kernels) — mirror of vllm/model_executor/layers/quantization/gptq.py (config :17-80, method :90-231).
2 / 3 / 4 / 8 bit (3-bit: 32 codes per 3 words, pack_factor = 32/3 as in the reference)."""
import enum
from fractions import Fraction
from enum import Enum
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import LinearMethodBase, QuantizationConfig, set_weight_attrs


class GPTQConfig(QuantizationConfig):

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, lm_head_quantized: bool = False) -> None:
        self.weight_bits, self.group_size, self.desc_act = weight_bits, group_size, desc_act
        self.lm_head_quantized = lm_head_quantized
        if self.weight_bits not in [2, 3, 4, 8]:
            raise ValueError("Currently, only 2/3/4/8-bit weight quantization is supported for GPTQ, "
                             f"but got {self.weight_bits} bits.")
        self.pack_factor = Fraction(32, self.weight_bits)  # 3-bit: 32/3 codes per word (gptq.py:36)

    def __repr__(self) -> str:
        return (f"GPTQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, desc_act={self.desc_act}),"
                f"lm_head_quantized={self.lm_head_quantized}")

    @classmethod
    def get_name(cls) -> str:
        return "gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half]

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]), cls.get_from_keys_or(config, ["lm_head"], default=False))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQLinearMethod"]:
        return GPTQLinearMethod(self)


class ExllamaState(Enum):
    UNUSED = enum.auto()
    UNINITIALIZED = enum.auto()
    READY = enum.auto()


class GPTQLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        if cfg.group_size != -1 and input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        out_pp = sum(output_partition_sizes)
        if out_pp % cfg.pack_factor.numerator != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        exllama_state = ExllamaState.UNINITIALIZED
        scale_and_zero_size = input_size // group_size
        scale_and_zero_input_dim = None
        if input_size != input_size_per_partition and cfg.group_size != -1:
            if cfg.desc_act:
                exllama_state = ExllamaState.UNUSED  # act-order + row-parallel: rows cannot be re-sorted per shard
            else:
                scale_and_zero_size = input_size_per_partition // group_size
                scale_and_zero_input_dim = 0
        qweight = Parameter(torch.empty(input_size_per_partition // cfg.pack_factor, out_pp, dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 0, "pack_factor": cfg.pack_factor})
        g_idx = Parameter(torch.tensor([i // cfg.group_size if cfg.group_size != -1 else 0 for i in range(input_size_per_partition)],
                                       dtype=torch.int32), requires_grad=False)
        set_weight_attrs(g_idx, {"input_dim": 0, "ignore_warning": True})
        qzeros = Parameter(torch.empty(scale_and_zero_size, out_pp // cfg.pack_factor, dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qzeros, {"input_dim": scale_and_zero_input_dim, "output_dim": 1, "packed_dim": 1,
                                  "pack_factor": cfg.pack_factor})
        scales = Parameter(torch.empty(scale_and_zero_size, out_pp, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": scale_and_zero_input_dim, "output_dim": 1})
        for name, prm in (("qweight", qweight), ("g_idx", g_idx), ("qzeros", qzeros), ("scales", scales)):
            layer.register_parameter(name, prm)
            set_weight_attrs(prm, extra_weight_attrs)
        layer.exllama_state = exllama_state

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        out_shape = x.shape[:-1] + (layer.qweight.shape[-1], )
        reshaped_x = x.reshape(-1, x.shape[-1])
        if layer.exllama_state == ExllamaState.UNINITIALIZED:  # one-time in-place shuffle (gptq.py:212-222)
            if self.quant_config.desc_act:
                layer.g_idx.data = torch.argsort(layer.g_idx).to(torch.int)
            else:
                layer.g_idx.data = torch.empty((0, ), device=layer.g_idx.device)
            layer.exllama_state = ExllamaState.READY
            ops.gptq_shuffle(layer.qweight, layer.g_idx, self.quant_config.weight_bits)
        output = ops.gptq_gemm(reshaped_x, layer.qweight, layer.qzeros, layer.scales, layer.g_idx,
                               layer.exllama_state == ExllamaState.READY, self.quant_config.weight_bits)
        if bias is not None:
            output.add_(bias)
        return output.reshape(out_shape)
