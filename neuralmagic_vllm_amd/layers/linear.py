"""Minimal TP-sharded linear layers driving a `QuantizeMethodBase` — the sharding contract of
vllm/model_executor/layers/linear.py (ColumnParallelLinear :232-346, RowParallelLinear :680-811) that the quantized
methods rely on: column-parallel layers split N (whole output channels, packed dims in packed units), row-parallel
layers split K and finish with ONE sum all-reduce of [M, hidden] (linear.py:791-793).
The reference's own layers are callers and stay the source of truth; these exist so that the TP path of this package
can run and be tested end to end."""
from typing import List, Optional

import torch
from torch.nn.parameter import Parameter

from neuralmagic_vllm_amd.distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                                              tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce)
from neuralmagic_vllm_amd.layers.quantization.base_config import QuantizationConfig


def divide(a: int, b: int) -> int:
    assert a % b == 0, f"{a} is not divisible by {b}"
    return a // b


class LinearBase(torch.nn.Module):

    def __init__(self, input_size: int, output_size: int, params_dtype: torch.dtype, quant_config: QuantizationConfig):
        super().__init__()
        self.input_size, self.output_size, self.params_dtype = input_size, output_size, params_dtype
        self.quant_method = quant_config.get_quant_method(self)


class ColumnParallelLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, quant_config: QuantizationConfig,
                 params_dtype: torch.dtype = torch.float16, gather_output: bool = False,
                 output_sizes: Optional[List[int]] = None):
        super().__init__(input_size, output_size, params_dtype, quant_config)
        tp = get_tensor_model_parallel_world_size()
        self.gather_output = gather_output
        self.output_size_per_partition = divide(output_size, tp)
        parts = [divide(s, tp) for s in output_sizes] if output_sizes else [self.output_size_per_partition]
        self.quant_method.create_weights(self, input_size, parts, input_size, output_size, params_dtype,
                                         weight_loader=self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        output_dim = getattr(param, "output_dim", None)
        if output_dim is not None:
            shard = param.data.shape[output_dim]  # packed dims are already in packed units
            loaded_weight = loaded_weight.narrow(output_dim, get_tensor_model_parallel_rank() * shard, shard)
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        assert param.data.shape == loaded_weight.shape, (param.data.shape, loaded_weight.shape)
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.quant_method.apply(self, x, None)
        return tensor_model_parallel_all_gather(out) if self.gather_output else out


class RowParallelLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, quant_config: QuantizationConfig,
                 params_dtype: torch.dtype = torch.float16, input_is_parallel: bool = True, reduce_results: bool = True):
        super().__init__(input_size, output_size, params_dtype, quant_config)
        tp = get_tensor_model_parallel_world_size()
        self.tp_size, self.input_is_parallel, self.reduce_results = tp, input_is_parallel, reduce_results
        self.input_size_per_partition = divide(input_size, tp)
        self.quant_method.create_weights(self, self.input_size_per_partition, [output_size], input_size, output_size,
                                         params_dtype, weight_loader=self.weight_loader)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor):
        input_dim = getattr(param, "input_dim", None)
        if input_dim is not None:
            shard = param.data.shape[input_dim]
            loaded_weight = loaded_weight.narrow(input_dim, get_tensor_model_parallel_rank() * shard, shard)
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        assert param.data.shape == loaded_weight.shape, (param.data.shape, loaded_weight.shape)
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.input_is_parallel:
            x = x.chunk(self.tp_size, dim=-1)[get_tensor_model_parallel_rank()].contiguous()
        out = self.quant_method.apply(self, x)
        if self.reduce_results and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
