"""INT8 W8A8 (compressed-tensors) — mirror of
vllm/model_executor/layers/quantization/compressed_tensors/schemes/compressed_tensors_w8a8.py:
int8 weights [N, K] with per-tensor or per-channel scales, static per-tensor or dynamic per-token activation scales."""
from typing import List, Optional

import torch
from torch.nn import Parameter

from neuralmagic_vllm_amd import _custom_ops as ops
from neuralmagic_vllm_amd.layers.quantization.base_config import set_weight_attrs


class CompressedTensorsW8A8:

    def __init__(self, strategy: str = "tensor", is_static_input_scheme: bool = False):
        assert strategy in ("tensor", "channel")
        self.strategy = strategy
        self.is_static_input_scheme = is_static_input_scheme

    def create_weights(self, layer: torch.nn.Module, output_partition_sizes: List[int], input_size_per_partition: int,
                       params_dtype: torch.dtype, weight_loader=None, **kwargs):
        out_pp = sum(output_partition_sizes)
        layer.logical_widths = output_partition_sizes
        weight = Parameter(torch.empty(out_pp, input_size_per_partition, dtype=torch.int8), requires_grad=False)
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "weight_loader": weight_loader})
        n_scale = out_pp if self.strategy == "channel" else len(output_partition_sizes)
        weight_scale = Parameter(torch.empty((n_scale, 1) if self.strategy == "channel" else (n_scale, ), dtype=torch.float32),
                                 requires_grad=False)
        layer.register_parameter("weight_scale", weight_scale)
        set_weight_attrs(weight_scale, {"weight_loader": weight_loader})
        if self.is_static_input_scheme:
            input_scale = Parameter(torch.empty(1, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("input_scale", input_scale)
            set_weight_attrs(input_scale, {"weight_loader": weight_loader, "ignore_warning": True})

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        layer.weight = Parameter(layer.weight.t(), requires_grad=False)  # column-major operand of scaled_mm
        if self.strategy == "tensor" and len(layer.logical_widths) > 1:
            # one scale per fused shard -> per-channel vector (compressed_tensors_w8a8.py:40-60)
            ws = torch.cat([layer.weight_scale[i].expand(w) for i, w in enumerate(layer.logical_widths)]).reshape(-1, 1)
            layer.weight_scale = Parameter(ws.contiguous(), requires_grad=False)
        if not self.is_static_input_scheme:
            layer.input_scale = None

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None):
        x_2d = x.reshape(-1, x.shape[-1])
        x_q, x_scale = ops.scaled_int8_quant(x_2d.contiguous(), layer.input_scale)
        out = ops.cutlass_scaled_mm(x_q, layer.weight, scale_a=x_scale, scale_b=layer.weight_scale.reshape(-1),
                                    out_dtype=x.dtype, bias=bias)
        return out.reshape(x.shape[:-1] + (out.shape[-1], ))
