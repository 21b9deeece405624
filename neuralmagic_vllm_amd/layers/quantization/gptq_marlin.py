"""GPTQ checkpoints served through the Marlin-format int4/int8 GEMM — mirror of
vllm/model_executor/layers/quantization/gptq_marlin.py (config :58-190, method :192-466).

Checkpoint parameters keep the reference's names, shapes and loader attributes (qweight [K/pf, N] int32 packed on
dim 0, g_idx [K], scales [K/g, N], qzeros on the meta device); the first `apply` repacks to the Marlin layout with
`ops.gptq_marlin_repack` and permutes the scales, then every call is one `ops.gptq_marlin_gemm`."""
import enum
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import (LinearMethodBase, QuantizationConfig,
                                                                  replace_tensor, set_weight_attrs)

GPTQ_MARLIN_TILE = 16
GPTQ_MARLIN_MIN_THREAD_N = 64
GPTQ_MARLIN_MIN_THREAD_K = 128
GPTQ_MARLIN_MAX_PARALLEL = 16
GPTQ_MARLIN_SUPPORTED_NUM_BITS = [4, 8]
GPTQ_MARLIN_SUPPORTED_GROUP_SIZES = [-1, 32, 64, 128]
GPTQ_MARLIN_SUPPORTED_SYM = [True]

_SCALE_PERM = [i + 8 * j for i in range(8) for j in range(8)]
_SCALE_PERM_SINGLE = [2 * i + j for i in range(4) for j in (0, 1, 8, 9, 16, 17, 24, 25)]


def get_pack_factor(num_bits: int) -> int:
    assert num_bits in GPTQ_MARLIN_SUPPORTED_NUM_BITS, f"Unsupported num_bits = {num_bits}"
    return 32 // num_bits


def marlin_permute_scales(s: torch.Tensor, size_k: int, size_n: int, group_size: int, num_bits: int) -> torch.Tensor:
    """gptq_marlin.py:47-56 — grouped scales: 8x8 transpose inside every 64 columns; channel-wise: the 32-column
    pattern of the single-row permutation."""
    if group_size < size_k and group_size != -1:
        s = s.reshape((-1, len(_SCALE_PERM)))[:, _SCALE_PERM]
    else:
        s = s.reshape((-1, len(_SCALE_PERM_SINGLE)))[:, _SCALE_PERM_SINGLE]
    return s.reshape((-1, size_n)).contiguous()


class GPTQMarlinState(enum.Enum):
    REPACK = enum.auto()
    READY = enum.auto()


class GPTQMarlinConfig(QuantizationConfig):

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, is_sym: bool,
                 lm_head_quantized: bool = False) -> None:
        if desc_act and group_size == -1:
            desc_act = False  # one group per channel: act-order is a no-op
        self.weight_bits, self.group_size, self.desc_act = weight_bits, group_size, desc_act
        self.is_sym, self.lm_head_quantized = is_sym, lm_head_quantized
        if weight_bits not in GPTQ_MARLIN_SUPPORTED_NUM_BITS:
            raise ValueError(f"Marlin does not support weight_bits = {weight_bits}. "
                             f"Only weight_bits = {GPTQ_MARLIN_SUPPORTED_NUM_BITS} are supported.")
        if group_size not in GPTQ_MARLIN_SUPPORTED_GROUP_SIZES:
            raise ValueError(f"Marlin does not support group_size = {group_size}. "
                             f"Only group_sizes = {GPTQ_MARLIN_SUPPORTED_GROUP_SIZES} are supported.")
        if is_sym not in GPTQ_MARLIN_SUPPORTED_SYM:
            raise ValueError(f"Marlin does not support is_sym = {is_sym}. Only sym = {GPTQ_MARLIN_SUPPORTED_SYM} are supported.")
        self.pack_factor = get_pack_factor(weight_bits)
        self.tile_size = GPTQ_MARLIN_TILE
        self.min_thread_n = GPTQ_MARLIN_MIN_THREAD_N
        self.min_thread_k = GPTQ_MARLIN_MIN_THREAD_K
        self.max_parallel = GPTQ_MARLIN_MAX_PARALLEL

    def __repr__(self) -> str:
        return (f"GPTQMarlinConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}, lm_head_quantized={self.lm_head_quantized})")

    @classmethod
    def get_name(cls) -> str:
        return "gptq_marlin"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQMarlinConfig":
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]), cls.get_from_keys(config, ["sym"]),
                   cls.get_from_keys_or(config, ["lm_head"], default=False))

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        # gptq_marlin.py:131-149: a plain "gptq" checkpoint is upgraded to gptq_marlin when compatible
        if cls.is_marlin_compatible(hf_quant_cfg) and user_quant in (None, "marlin", "gptq_marlin"):
            return cls.get_name()
        return None

    @classmethod
    def is_marlin_compatible(cls, quant_config: Dict[str, Any]) -> bool:
        # gptq_marlin.py:158-190 minus the CUDA capability gate (gfx950 runs every supported combination)
        num_bits, group_size = quant_config.get("bits"), quant_config.get("group_size")
        sym, desc_act = quant_config.get("sym"), quant_config.get("desc_act")
        if num_bits is None or group_size is None or sym is None or desc_act is None:
            return False
        return (num_bits in GPTQ_MARLIN_SUPPORTED_NUM_BITS and group_size in GPTQ_MARLIN_SUPPORTED_GROUP_SIZES
                and sym in GPTQ_MARLIN_SUPPORTED_SYM)

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["GPTQMarlinLinearMethod"]:
        return GPTQMarlinLinearMethod(self)


class GPTQMarlinLinearMethod(LinearMethodBase):

    def __init__(self, quant_config: GPTQMarlinConfig) -> None:
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs) -> None:
        cfg = self.quant_config
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        if params_dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"The params dtype must be float16 or bfloat16, but got {params_dtype}")
        out_pp = sum(output_partition_sizes)
        if out_pp % cfg.min_thread_n != 0:
            raise ValueError(f"Weight output_size_per_partition = {out_pp} is not divisible by "
                             f" min_thread_n = {cfg.min_thread_n}.")
        if input_size_per_partition % cfg.min_thread_k != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is not divisible "
                             f"by min_thread_k = {cfg.min_thread_k}.")
        if group_size < input_size and input_size_per_partition % group_size != 0:
            raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition}"
                             f" is not divisible by group_size = {group_size}.")

        # sharding of scales / zero points along K (row-parallel layers) — gptq_marlin.py:245-270
        scales_and_zp_size = input_size // group_size
        scales_and_zp_input_dim = None
        if cfg.desc_act:
            assert cfg.group_size != -1
            is_k_full = input_size_per_partition == input_size
        else:
            is_k_full = True
            if input_size != input_size_per_partition and cfg.group_size != -1:
                scales_and_zp_size = input_size_per_partition // group_size
                scales_and_zp_input_dim = 0

        qweight = Parameter(torch.empty(input_size_per_partition // cfg.pack_factor, out_pp, dtype=torch.int32),
                            requires_grad=False)
        set_weight_attrs(qweight, {**extra_weight_attrs, "input_dim": 0, "output_dim": 1, "packed_dim": 0,
                                   "pack_factor": cfg.pack_factor})
        g_idx = Parameter(torch.empty(input_size_per_partition, dtype=torch.int32), requires_grad=False)
        set_weight_attrs(g_idx, {**extra_weight_attrs, "input_dim": 0, "ignore_warning": True})
        scales = Parameter(torch.empty(scales_and_zp_size, out_pp, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {**extra_weight_attrs, "input_dim": scales_and_zp_input_dim, "output_dim": 1})
        qzeros = Parameter(torch.empty(scales_and_zp_size, out_pp // cfg.pack_factor, dtype=torch.int32, device="meta"),
                           requires_grad=False)
        set_weight_attrs(qzeros, {**extra_weight_attrs, "input_dim": scales_and_zp_input_dim, "output_dim": 1,
                                  "packed_dim": 1, "pack_factor": cfg.pack_factor})

        layer.register_parameter("qweight", qweight)
        layer.register_parameter("g_idx", g_idx)
        layer.register_parameter("scales", scales)
        layer.register_parameter("qzeros", qzeros)
        layer.g_idx_sort_indices = torch.empty(g_idx.shape, dtype=torch.int32)
        layer.workspace = torch.zeros((out_pp // cfg.min_thread_n) * cfg.max_parallel, dtype=torch.int)
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = out_pp
        layer.input_size = input_size
        layer.is_k_full = is_k_full
        layer.marlin_state = GPTQMarlinState.REPACK

    def _repack(self, layer: torch.nn.Module) -> None:
        cfg = self.quant_config
        dev = layer.qweight.device
        if cfg.desc_act:
            sort_idx = torch.argsort(layer.g_idx).to(torch.int)
            sorted_g_idx = layer.g_idx[sort_idx]
            replace_tensor(layer, "g_idx", sorted_g_idx)
            layer.g_idx_sort_indices = sort_idx.to(dev)
        else:
            layer.g_idx = Parameter(torch.empty(0, dtype=torch.int, device=dev), requires_grad=False)
            layer.g_idx_sort_indices = torch.empty(0, dtype=torch.int, device=dev)
        layer.workspace = layer.workspace.to(dev)
        marlin_qweight = ops.gptq_marlin_repack(layer.qweight, layer.g_idx_sort_indices, layer.input_size_per_partition,
                                                layer.output_size_per_partition, cfg.weight_bits)
        replace_tensor(layer, "qweight", marlin_qweight)
        scales_size_k = layer.input_size if cfg.desc_act else layer.input_size_per_partition
        replace_tensor(layer, "scales", marlin_permute_scales(layer.scales, scales_size_k, layer.output_size_per_partition,
                                                              cfg.group_size, cfg.weight_bits))

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        reshaped_x = x.reshape(-1, x.shape[-1])
        out_shape = x.shape[:-1] + (layer.output_size_per_partition, )
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            self._repack(layer)
        output = ops.gptq_marlin_gemm(reshaped_x, layer.qweight, layer.scales, layer.g_idx, layer.g_idx_sort_indices,
                                      layer.workspace, self.quant_config.weight_bits, reshaped_x.shape[0],
                                      layer.output_size_per_partition, layer.input_size_per_partition, layer.is_k_full)
        if bias is not None:
            output.add_(bias)
        return output.reshape(out_shape)

    def apply_silu_and_mul(self, layer: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
        """SiluAndMul(apply(layer, x)) for a merged gate | up projection without bias as ONE op (extension; the reference's
        LlamaMLP.forward runs gate_up_proj, then act_fn: models/llama.py:79-83). Same values as the two calls."""
        reshaped_x = x.reshape(-1, x.shape[-1])
        if layer.marlin_state == GPTQMarlinState.REPACK:
            layer.marlin_state = GPTQMarlinState.READY
            self._repack(layer)
        act = ops.gptq_marlin_gemm_silu_and_mul(reshaped_x, layer.qweight, layer.scales, layer.g_idx, layer.g_idx_sort_indices,
                                                layer.workspace, self.quant_config.weight_bits, reshaped_x.shape[0],
                                                layer.output_size_per_partition, layer.input_size_per_partition,
                                                layer.is_k_full)
        return act.reshape(x.shape[:-1] + (layer.output_size_per_partition // 2, ))
