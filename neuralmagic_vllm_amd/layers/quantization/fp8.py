"""FP8 (W8A8, per-tensor scales; fp8 KV-cache scale) — mirror of vllm/model_executor/layers/quantization/fp8.py
(config :35-89, linear method :92-379, KV-cache method :563-598). On gfx950 the fp8 x fp8 MFMA path is native, so
`cutlass_scaled_mm_supports_fp8` is always true and the W8A16 Marlin fallback (fp8.py:116-118) is opt-in only."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn import Module
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import (LinearMethodBase, QuantizationConfig,
                                                                  QuantizeMethodBase, set_weight_attrs)

ACTIVATION_SCHEMES = ["static", "dynamic"]


class Fp8Config(QuantizationConfig):

    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic") -> None:
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.activation_scheme = activation_scheme

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized=("fp8" in quant_method),
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]))

    def get_quant_method(self, layer: torch.nn.Module) -> Optional["QuantizeMethodBase"]:
        return Fp8LinearMethod(self)


class Fp8LinearMethod(LinearMethodBase):
    """Per-tensor fp8 weights (stored [N, K] e4m3fn, used column-major as `weight.t()`), static or dynamic per-tensor
    activation scale. A fused module (QKV, gate_up) loads one scale per logical weight; they are collapsed to the max
    with a requantisation (fp8.py:239-276)."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.cutlass_fp8_supported = ops.cutlass_scaled_mm_supports_fp8(95)

    def _create_scale_param(self, scale_name: str, layer: torch.nn.Module, output_partition_sizes: List[int],
                            **extra_weight_attrs) -> None:
        scale = Parameter(torch.empty(len(output_partition_sizes), dtype=torch.float32), requires_grad=False)
        scale[:] = torch.finfo(torch.float8_e4m3fn).min
        layer.register_parameter(scale_name, scale)
        set_weight_attrs(scale, {**extra_weight_attrs, "needs_scalar_to_array": True})

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        out_pp = sum(output_partition_sizes)
        layer.process_after_load = True
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = out_pp
        layer.orig_dtype = params_dtype
        weight_dtype = torch.float8_e4m3fn if self.quant_config.is_checkpoint_fp8_serialized else params_dtype
        weight = Parameter(torch.empty(out_pp, input_size_per_partition, dtype=weight_dtype), requires_grad=False)
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, {**extra_weight_attrs, "input_dim": 1, "output_dim": 0})
        if self.quant_config.is_checkpoint_fp8_serialized:
            self._create_scale_param("weight_scale", layer, output_partition_sizes, **extra_weight_attrs)
            if self.quant_config.activation_scheme == "static":
                self._create_scale_param("input_scale", layer, output_partition_sizes, **extra_weight_attrs)

    def process_weights_after_loading(self, layer: Module) -> None:
        if not getattr(layer, "process_after_load", False):
            return
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # fp16 / bf16 checkpoint: quantise the weight once, dynamically, per tensor (fp8.py:226-237)
            qweight, weight_scale = ops.scaled_fp8_quant(layer.weight, scale=None)
            layer.weight = Parameter(qweight.t(), requires_grad=False)
            layer.weight_scale = Parameter(weight_scale, requires_grad=False)
            layer.logical_widths = None
            layer.input_scale = None
            return
        # fp8 checkpoint with one scale per logical weight: requantise every shard to the max scale
        max_w_scale = layer.weight_scale.max()
        unfused = len(layer.logical_widths) > 1 and bool((layer.weight_scale != layer.weight_scale[0]).any())
        if unfused:
            start = 0
            for idx, width in enumerate(layer.logical_widths):
                end = start + width
                deq = layer.weight[start:end, :].to(torch.float32) * layer.weight_scale[idx]
                layer.weight[start:end, :] = ops.scaled_fp8_quant(deq.to(layer.orig_dtype), max_w_scale.reshape(1))[0]
                start = end
        layer.weight_scale = Parameter(max_w_scale.reshape(1), requires_grad=False)
        layer.weight = Parameter(layer.weight.t(), requires_grad=False)
        if self.quant_config.activation_scheme == "dynamic":
            layer.input_scale = None
        else:
            layer.input_scale = Parameter(layer.input_scale.max().reshape(1), requires_grad=False)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x_2d = x.reshape(-1, x.shape[-1])
        # static: input_scale is a scalar tensor; dynamic: computed from x (fp8.py:343-359)
        qinput, x_scale = ops.scaled_fp8_quant(x_2d, layer.input_scale)
        out = ops.cutlass_scaled_mm(qinput, layer.weight, scale_a=x_scale, scale_b=layer.weight_scale, out_dtype=x.dtype,
                                    bias=bias)
        return out.reshape(x.shape[:-1] + (out.shape[-1], ))


class Fp8MoEMethod(QuantizeMethodBase):
    """MoE method for FP8 — mirror of fp8.py:382-560 (Fp8MoEMethod): fp8 or fp16/bf16 checkpoints, per-expert weight scales
    (one for the merged w13 after loading), one static or dynamic activation scale. `apply` runs fused_moe on the
    MI355X ops (layers/fused_moe.py)."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: Module, num_experts: int, hidden_size: int, intermediate_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        layer.process_after_load = True
        layer.num_experts = num_experts
        layer.intermediate_size_per_partition = intermediate_size
        if self.quant_config.is_checkpoint_fp8_serialized:
            params_dtype = torch.float8_e4m3fn
        w13_weight = Parameter(torch.empty(num_experts, 2 * intermediate_size, hidden_size, dtype=params_dtype), requires_grad=False)
        layer.register_parameter("w13_weight", w13_weight)
        set_weight_attrs(w13_weight, extra_weight_attrs)
        w2_weight = Parameter(torch.empty(num_experts, hidden_size, intermediate_size, dtype=params_dtype), requires_grad=False)
        layer.register_parameter("w2_weight", w2_weight)
        set_weight_attrs(w2_weight, extra_weight_attrs)
        # two scales for w1 and w3; combined into one after loading (fp8.py:424-441)
        w13_scale = Parameter(torch.ones(num_experts, 2, dtype=torch.float32), requires_grad=False)
        layer.register_parameter("w13_scale", w13_scale)
        w2_scale = Parameter(torch.ones(num_experts, dtype=torch.float32), requires_grad=False)
        layer.register_parameter("w2_scale", w2_scale)
        if self.quant_config.is_checkpoint_fp8_serialized:
            set_weight_attrs(w13_scale, extra_weight_attrs)
            set_weight_attrs(w2_scale, extra_weight_attrs)
        if self.quant_config.activation_scheme == "static":
            if not self.quant_config.is_checkpoint_fp8_serialized:
                raise ValueError("Found static activation scheme for checkpoint that was not serialized fp8.")
            a13_scale = Parameter(torch.ones(num_experts, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("a13_scale", a13_scale)
            set_weight_attrs(a13_scale, extra_weight_attrs)
            a2_scale = Parameter(torch.ones(num_experts, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("a2_scale", a2_scale)
            set_weight_attrs(a2_scale, extra_weight_attrs)
        else:
            layer.a13_scale = None
            layer.a2_scale = None

    def process_weights_after_loading(self, layer: Module) -> None:
        if not getattr(layer, "process_after_load", False):
            return
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # fp16 / bf16 checkpoint: quantise every expert in place, one scale for the merged w13 (fp8.py:470-494)
            w13 = torch.empty_like(layer.w13_weight.data, dtype=torch.float8_e4m3fn)
            w2 = torch.empty_like(layer.w2_weight.data, dtype=torch.float8_e4m3fn)
            w13_scale = torch.ones(layer.num_experts, dtype=torch.float32, device=w13.device)
            w2_scale = torch.ones(layer.num_experts, dtype=torch.float32, device=w13.device)
            for ex in range(layer.num_experts):
                w13[ex], s13 = ops.scaled_fp8_quant(layer.w13_weight.data[ex])
                w2[ex], s2 = ops.scaled_fp8_quant(layer.w2_weight.data[ex])
                w13_scale[ex], w2_scale[ex] = s13[0], s2[0]
            layer.w13_weight = Parameter(w13, requires_grad=False)
            layer.w2_weight = Parameter(w2, requires_grad=False)
            layer.w13_scale = Parameter(w13_scale, requires_grad=False)
            layer.w2_scale = Parameter(w2_scale, requires_grad=False)
            return
        # fp8 checkpoint: one activation scale (max over the experts) and one w13 scale per expert (fp8.py:499-540)
        if self.quant_config.activation_scheme == "static":
            if layer.a13_scale is None or layer.a2_scale is None:
                raise ValueError("QuantConfig has static quantization, but found activation scales are None.")
            layer.a13_scale = Parameter(layer.a13_scale.max().reshape(1), requires_grad=False)
            layer.a2_scale = Parameter(layer.a2_scale.max().reshape(1), requires_grad=False)
        shard = layer.intermediate_size_per_partition
        max_w13 = layer.w13_scale.max(dim=1).values
        for ex in range(layer.num_experts):
            for sid in range(2):
                dq = per_tensor_dequantize(layer.w13_weight[ex][sid * shard:(sid + 1) * shard, :], layer.w13_scale[ex][sid])
                layer.w13_weight[ex][sid * shard:(sid + 1) * shard, :] = per_tensor_quantize(dq, max_w13[ex])
        layer.w13_scale = Parameter(max_w13, requires_grad=False)

    def apply(self, layer: Module, x: torch.Tensor, router_logits: torch.Tensor, top_k: int, renormalize: bool = True) -> torch.Tensor:
        from neuralmagic_vllm_amd.layers.fused_moe import fused_moe
        return fused_moe(x, layer.w13_weight, layer.w2_weight, router_logits, top_k, renormalize=renormalize, inplace=True,
                         use_fp8=True, w1_scale=layer.w13_scale, w2_scale=layer.w2_scale, a1_scale=layer.a13_scale,
                         a2_scale=layer.a2_scale)


def per_tensor_quantize(tensor: torch.Tensor, inv_scale) -> torch.Tensor:
    """fp8.py:612-617"""
    finfo = torch.finfo(torch.float8_e4m3fn)
    qweight = (tensor / inv_scale).clamp(min=finfo.min, max=finfo.max)
    return qweight.to(torch.float8_e4m3fn)


def per_tensor_dequantize(tensor: torch.Tensor, inv_scale) -> torch.Tensor:
    """fp8.py:620-624"""
    return tensor.to(torch.float16) * inv_scale


class Fp8KVCacheMethod(QuantizeMethodBase):
    """fp8 KV-cache scaling factor loaded from the checkpoint or a JSON file (fp8.py:563-598); reaches the kernels as the
    scalar `kv_scale` of reshape_and_cache / paged_attention."""

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module):
        layer.kv_scale = Parameter(torch.tensor(1.0), requires_grad=False)

    def apply(self, layer: torch.nn.Module) -> torch.Tensor:
        raise RuntimeError("Fp8KVCacheMethod.apply should not be called.")

    def process_weights_after_loading(self, layer: Module) -> None:
        if getattr(layer, "kv_cache_dtype", "auto") == "fp8":
            kv_scale = layer.kv_scale.to("cpu").tolist()
            if not isinstance(kv_scale, float):
                raise ValueError("Only support per-tensor scaling factor for fp8 KV cache")
            layer._kv_scale = kv_scale
        del layer.kv_scale
