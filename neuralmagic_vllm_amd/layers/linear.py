"""Minimal TP-sharded linear layers driving a `QuantizeMethodBase` — the sharding contract of
vllm/model_executor/layers/linear.py (ColumnParallelLinear :232-346, RowParallelLinear :680-811) that the quantized
methods rely on: column-parallel layers split N (whole output channels, packed dims in packed units), row-parallel
layers split K and finish with ONE sum all-reduce of [M, hidden] (linear.py:791-793).
The reference's own layers are callers and stay the source of truth; these exist so that the TP path of this package
can run and be tested end to end."""
from typing import List, Optional, Tuple, Union

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
                 output_sizes: Optional[List[int]] = None, **weight_attrs):
        super().__init__(input_size, output_size, params_dtype, quant_config)
        tp = get_tensor_model_parallel_world_size()
        self.gather_output = gather_output
        self.output_size_per_partition = divide(output_size, tp)
        parts = [divide(s, tp) for s in output_sizes] if output_sizes else [self.output_size_per_partition]
        # weight_attrs: extra create_weights keywords (e.g. device= for the methods that allocate on "cuda" by default)
        self.quant_method.create_weights(self, input_size, parts, input_size, output_size, params_dtype,
                                         weight_loader=self.weight_loader, **weight_attrs)

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


# ---- fused layers whose checkpoint stores the parts separately (q/k/v, gate/up) --------------------------------------
def adjust_marlin_shard(param: Parameter, shard_size: int, shard_offset: int) -> Tuple[int, int]:
    """Marlin-format parameters pack `marlin_tile_size` k-rows into the column dimension (linear.py:23-28)."""
    tile = getattr(param, "marlin_tile_size", None)
    return (shard_size, shard_offset) if tile is None else (shard_size * tile, shard_offset * tile)


def adjust_scalar_to_fused_array(param_data: torch.Tensor, loaded_weight: torch.Tensor,
                                 shard_id: Union[str, int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """A fused module keeps one per-tensor scale per logical matrix; the checkpoint holds a scalar (AutoFP8) or a
    one-element tensor (compressed-tensors) per part (linear.py:46-66)."""
    idx = {"q": 0, "k": 1, "v": 2}[shard_id] if isinstance(shard_id, str) else shard_id
    if not isinstance(idx, int):
        raise ValueError(f"Unknown Shard Id {shard_id}")
    if loaded_weight.dim() != 0:
        assert loaded_weight.shape[0] == 1
        loaded_weight = loaded_weight[0]
    return param_data[idx], loaded_weight


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Column-parallel layer whose output is the concatenation of several checkpoint matrices (gate_proj | up_proj):
    linear.py:336-470. `weight_loader(param, loaded_weight, loaded_shard_id)` places part `loaded_shard_id` of this
    rank's shard; offsets along a packed output dimension are divided by the pack factor (and scaled by the Marlin tile
    for Marlin-format tensors)."""

    def __init__(self, input_size: int, output_sizes: List[int], quant_config: QuantizationConfig,
                 params_dtype: torch.dtype = torch.float16, gather_output: bool = False, **weight_attrs):
        self.output_sizes = list(output_sizes)
        tp = get_tensor_model_parallel_world_size()
        assert all(s % tp == 0 for s in output_sizes)
        super().__init__(input_size, sum(output_sizes), quant_config, params_dtype, gather_output, output_sizes=output_sizes,
                         **weight_attrs)

    def _shards(self) -> List[Tuple[Union[str, int], int, int]]:
        out, off = [], 0
        for i, size in enumerate(self.output_sizes):
            out.append((i, off, size))
            off += size
        return out

    def _rank_slice(self, shard_id) -> Tuple[int, int, int]:
        """(offset of the part inside this rank's parameter, size of the rank's share, index of that share inside the
        checkpoint tensor), in unpacked output channels."""
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        return sum(self.output_sizes[:shard_id]) // tp, self.output_sizes[shard_id] // tp, rank

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor, loaded_shard_id=None):
        param_data = param.data
        output_dim = getattr(param, "output_dim", None)
        needs_scalar_to_array = getattr(param, "needs_scalar_to_array", False)
        packed = getattr(param, "packed_dim", None) == output_dim and output_dim is not None
        if loaded_shard_id is None:
            # the checkpoint tensor is already fused: split it and load part by part
            if output_dim is None:
                if needs_scalar_to_array:
                    param_data, loaded_weight = adjust_scalar_to_fused_array(param_data, loaded_weight, 0)
                assert param_data.shape == loaded_weight.shape, (param_data.shape, loaded_weight.shape)
                param_data.copy_(loaded_weight)
                return
            for shard_id, off, size in self._shards():
                if packed:
                    size, off = adjust_marlin_shard(param, size // param.pack_factor, off // param.pack_factor)
                self.weight_loader(param, loaded_weight.narrow(output_dim, off, size), shard_id)
            return
        if output_dim is not None:
            off, size, idx = self._rank_slice(loaded_shard_id)
            if packed:
                size, off = adjust_marlin_shard(param, size // param.pack_factor, off // param.pack_factor)
            param_data = param_data.narrow(output_dim, off, size)
            loaded_weight = loaded_weight.narrow(output_dim, idx * size, size)
        elif needs_scalar_to_array:
            param_data, loaded_weight = adjust_scalar_to_fused_array(param_data, loaded_weight, loaded_shard_id)
        # else: no output dimension (g_idx, channel-less metadata): the same tensor on every rank and for every part
        input_dim = getattr(param, "input_dim", None)
        if input_dim is not None and loaded_weight.shape[input_dim] != param_data.shape[input_dim]:
            raise ValueError("column-parallel layers do not shard the input dimension")
        if loaded_weight.dim() == 0 and param_data.dim() == 1 and param_data.numel() == 1:
            loaded_weight = loaded_weight.reshape(1)
        assert param_data.shape == loaded_weight.shape, (param_data.shape, loaded_weight.shape)
        param_data.copy_(loaded_weight)


class QKVParallelLinear(MergedColumnParallelLinear):
    """Fused q | k | v projection (linear.py:473-677): heads are divided over the TP ranks; when there are fewer kv heads
    than ranks each kv head is replicated on tp / kv_heads ranks (config.py:396-404)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: Optional[int],
                 quant_config: QuantizationConfig, params_dtype: torch.dtype = torch.float16):
        self.hidden_size, self.head_size = hidden_size, head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = divide(self.total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, divide(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads, self.num_kv_head_replicas = divide(self.total_num_kv_heads, tp), 1
        sizes = [self.num_heads * head_size * tp, self.num_kv_heads * head_size * tp, self.num_kv_heads * head_size * tp]
        super().__init__(hidden_size, sizes, quant_config, params_dtype)

    def _shards(self):
        q, kv = self.total_num_heads * self.head_size, self.total_num_kv_heads * self.head_size
        return [("q", 0, q), ("k", q, kv), ("v", q + kv, kv)]

    def _rank_slice(self, shard_id):
        assert shard_id in ("q", "k", "v")
        rank = get_tensor_model_parallel_rank()
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        if shard_id == "q":
            return 0, q, rank
        return (q if shard_id == "k" else q + kv), kv, rank // self.num_kv_head_replicas
