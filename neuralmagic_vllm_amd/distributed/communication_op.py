"""vllm/distributed/communication_op.py:9-21 — the two collectives the TP-sharded linears call."""
import torch

from neuralmagic_vllm_amd.distributed.parallel_state import get_tp_group, get_tensor_model_parallel_world_size


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """All-reduce the input tensor across the tensor-parallel group (RCCL ring / tree over xGMI)."""
    if get_tensor_model_parallel_world_size() == 1:
        return input_
    return get_tp_group().all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    if get_tensor_model_parallel_world_size() == 1:
        return input_
    return get_tp_group().all_gather(input_, dim)
