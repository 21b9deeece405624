"""Checkpoints stored directly in the Marlin layout — mirror of vllm/model_executor/layers/quantization/marlin.py
(config :18-111, method :114-256): parameters `B` [K/16, N*16/8] int32, `s` [groups, N] fp16, `workspace`."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import LinearMethodBase, QuantizationConfig, set_weight_attrs


class MarlinConfig(QuantizationConfig):

    def __init__(self, group_size: int, lm_head_quantized: bool = False) -> None:
        self.group_size = group_size
        self.lm_head_quantized = lm_head_quantized
        if self.group_size != 128 and self.group_size != -1:
            raise ValueError("Currently, only group size 128 and -1 (channelwise) is supported for Marlin, "
                             f"but got group_size of {self.group_size}")
        self.pack_factor = 32 // 4   # 4-bit weights in int32
        self.tile_size = 16
        self.min_n_threads = 64
        self.min_k_threads = 128
        self.max_parallel = 16
        self.perm_len = 1024

    def __repr__(self) -> str:
        return f"MarlinConfig(group_size={self.group_size}, lm_head_quantized={self.lm_head_quantized})"

    @classmethod
    def get_name(cls) -> str:
        return "marlin"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half]

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MarlinConfig":
        return cls(cls.get_from_keys(config, ["group_size"]), cls.get_from_keys_or(config, ["lm_head"], default=False))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["MarlinLinearMethod"]:
        return MarlinLinearMethod(self)


class MarlinLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: MarlinConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        if params_dtype != torch.float16:
            raise ValueError(f"The params dtype must be float16, but got {params_dtype}")
        out_pp = sum(output_partition_sizes)
        if out_pp % cfg.min_n_threads != 0:
            raise ValueError(f"Weight output_size_per_partition = {out_pp} is not divisible by min_n_threads = {cfg.min_n_threads}.")
        if out_pp % cfg.pack_factor != 0:
            raise ValueError(f"Weight output_size_per_partition = {out_pp} is not divisible by pack_factor = {cfg.pack_factor}.")
        if input_size_per_partition % cfg.min_k_threads != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is not divisible by "
                             f"min_k_threads = {cfg.min_k_threads}.")
        if cfg.group_size != -1 and input_size_per_partition % cfg.group_size != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is not divisible by "
                             f"group_size = {cfg.group_size}.")
        if out_pp % (cfg.perm_len // (cfg.tile_size**2)) != 0:
            raise ValueError("Each permutation group must reside on the same gpu")
        dev = extra_weight_attrs.pop("device", "cuda")
        qweight = Parameter(torch.empty(input_size_per_partition // cfg.tile_size, out_pp * cfg.tile_size // cfg.pack_factor,
                                        device=dev, dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": cfg.pack_factor,
                                   "marlin_tile_size": cfg.tile_size})
        input_groups = 1 if cfg.group_size == -1 else input_size_per_partition // cfg.group_size
        scales = Parameter(torch.empty(input_groups, out_pp, device=dev, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": None if input_groups == 1 else 0, "output_dim": 1})
        workspace = Parameter(torch.zeros((out_pp // cfg.min_n_threads) * cfg.max_parallel, device=dev, dtype=torch.int),
                              requires_grad=False)
        layer.register_parameter("B", qweight)
        set_weight_attrs(qweight, extra_weight_attrs)
        layer.register_parameter("s", scales)
        set_weight_attrs(scales, extra_weight_attrs)
        layer.register_parameter("workspace", workspace)
        set_weight_attrs(workspace, extra_weight_attrs)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x_2d = x.view(-1, x.shape[-1])
        out = ops.marlin_gemm(x_2d, layer.B, layer.s, layer.workspace, x_2d.shape[0], layer.s.shape[1], x_2d.shape[1])
        out = out.view(x.shape[:-1] + (out.shape[1], ))
        if bias is not None:
            out.add_(bias)
        return out
