"""2:4-sparse + int4/int8 checkpoints in the Marlin-24 layout — mirror of
vllm/model_executor/layers/quantization/gptq_marlin_24.py (config :25-120, method :123-291): parameters
`B_24` [K/16/2, N*16/pack] int32, `B_meta` [K/32, 2N] int16, `s` [groups, N] fp16, `workspace`."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import LinearMethodBase, QuantizationConfig, set_weight_attrs

GPTQ_MARLIN_24_TILE = 16
GPTQ_MARLIN_24_MIN_THREAD_N = 128
GPTQ_MARLIN_24_MIN_THREAD_K = 128
GPTQ_MARLIN_24_MAX_PARALLEL = 64
GPTQ_MARLIN_24_SUPPORTED_NUM_BITS = [4, 8]
GPTQ_MARLIN_24_SUPPORTED_GROUP_SIZES = [-1, 128]


class GPTQMarlin24Config(QuantizationConfig):

    def __init__(self, weight_bits: int, group_size: int) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        if self.weight_bits not in GPTQ_MARLIN_24_SUPPORTED_NUM_BITS:
            raise ValueError(f"Marlin_24 does not support weight_bits = {self.weight_bits}. "
                             f"Only weight_bits = {GPTQ_MARLIN_24_SUPPORTED_NUM_BITS} are supported.")
        if self.group_size not in GPTQ_MARLIN_24_SUPPORTED_GROUP_SIZES:
            raise ValueError(f"Marlin_24 does not support group_size = {self.group_size}. "
                             f"Only group_sizes = {GPTQ_MARLIN_24_SUPPORTED_GROUP_SIZES} are supported.")
        self.pack_factor = 32 // self.weight_bits
        self.tile_size = GPTQ_MARLIN_24_TILE
        self.min_n_threads = GPTQ_MARLIN_24_MIN_THREAD_N
        self.min_k_threads = GPTQ_MARLIN_24_MIN_THREAD_K
        self.max_parallel = GPTQ_MARLIN_24_MAX_PARALLEL
        self.perm_len = 1024

    def __repr__(self) -> str:
        return f"Marlin24Config(weight_bits={self.weight_bits}, group_size={self.group_size})"

    @classmethod
    def get_name(cls) -> str:
        return "gptq_marlin_24"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half]

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQMarlin24Config":
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]))

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        is_marlin_24_format = hf_quant_cfg.get("checkpoint_format") == "marlin_24"
        is_valid_user_quant = user_quant is None or user_quant == "gptq" or user_quant == "gptq_marlin_24"
        return cls.get_name() if (is_marlin_24_format and is_valid_user_quant) else None

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQMarlin24LinearMethod"]:
        return GPTQMarlin24LinearMethod(self)


class GPTQMarlin24LinearMethod(LinearMethodBase):

    def __init__(self, quant_config: GPTQMarlin24Config):
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
        qweight = Parameter(torch.empty(input_size_per_partition // cfg.tile_size // 2,
                                        out_pp * cfg.tile_size // cfg.pack_factor, device=dev, dtype=torch.int32),
                            requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": cfg.pack_factor,
                                   "marlin_tile_size": cfg.tile_size})
        meta = Parameter(torch.empty(input_size_per_partition // 8 // 2 // 2, out_pp * 2, device=dev, dtype=torch.int16),
                         requires_grad=False)
        set_weight_attrs(meta, {"input_dim": 0, "packed_dim": 1, "pack_factor": 1, "output_dim": 1, "marlin_tile_size": 2})
        input_groups = 1 if cfg.group_size == -1 else input_size_per_partition // cfg.group_size
        scales = Parameter(torch.empty(input_groups, out_pp, device=dev, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": None if input_groups == 1 else 0, "output_dim": 1})
        workspace = Parameter(torch.zeros((out_pp // cfg.min_n_threads) * cfg.max_parallel, device=dev, dtype=torch.int),
                              requires_grad=False)
        layer.register_parameter("B_24", qweight)
        set_weight_attrs(qweight, extra_weight_attrs)
        layer.register_parameter("B_meta", meta)
        set_weight_attrs(meta, extra_weight_attrs)
        layer.register_parameter("s", scales)
        set_weight_attrs(scales, extra_weight_attrs)
        layer.register_parameter("workspace", workspace)
        set_weight_attrs(workspace, extra_weight_attrs)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x_2d = x.view(-1, x.shape[-1])
        out = ops.gptq_marlin_24_gemm(x_2d, layer.B_24, layer.B_meta, layer.s, layer.workspace, self.quant_config.weight_bits,
                                      x_2d.shape[0], layer.s.shape[1], x_2d.shape[1])
        out = out.view(x.shape[:-1] + (out.shape[1], ))
        if bias is not None:
            out.add_(bias)
        return out
