"""Tensor-parallel group state — the slice of vllm/distributed/parallel_state.py the hot path touches
(GroupCoordinator.all_reduce / all_gather, :273-340; init_distributed_environment :780-820).

One process per GPU; the device group runs on `torch.distributed` backend "nccl", which IS RCCL on ROCm and rides the
xGMI mesh of an MI355X node. The reference's CUDA-IPC custom all-reduce is compiled out on ROCm
(csrc/torch_bindings.cpp:261); here RCCL carries every message, issued on the compute stream so that it is
hipGraph-capturable like the reference's pynccl path (device_communicators/pynccl.py:99-118). CPU tensors (tests) use a
gloo group."""
import os
from typing import Optional

import torch
import torch.distributed as dist


class GroupCoordinator:

    def __init__(self, ranks, local_rank: int, backend: str):
        self.ranks = list(ranks)
        self.world_size = len(self.ranks)
        self.rank = dist.get_rank()
        self.rank_in_group = self.ranks.index(self.rank)
        self.local_rank = local_rank
        self.device_group = dist.new_group(self.ranks, backend=backend)
        self.backend = backend
        # small decode messages: the one-shot xGMI all-reduce, attached to a CPU (gloo) group for the IPC-handle exchange
        # like the reference (parallel_state.py:166-179). Off unless NMX_CUSTOM_AR=1 (custom_all_reduce.py).
        self.ca_comm = None
        if backend == "nccl" and self.world_size > 1 and os.environ.get("NMX_CUSTOM_AR", "0") == "1":
            from neuralmagic_vllm_amd.distributed.custom_all_reduce import CustomAllreduce
            self.cpu_group = dist.new_group(self.ranks, backend="gloo")
            self.ca_comm = CustomAllreduce(self.cpu_group, torch.device("cuda", local_rank))

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the group (parallel_state.py:273-293). Bypassed for world size 1."""
        if self.world_size == 1:
            return input_
        if self.ca_comm is not None:
            out = self.ca_comm.custom_all_reduce(input_)  # None: message not served by the one-shot kernel
            if out is not None:
                return out
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def all_gather(self, input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """Concatenate along `dim` over the group (parallel_state.py:295-321)."""
        if self.world_size == 1:
            return input_
        if dim < 0:
            dim += input_.dim()
        sizes0 = list(input_.size())
        out = torch.empty([self.world_size * sizes0[0]] + sizes0[1:], dtype=input_.dtype, device=input_.device)
        dist.all_gather_into_tensor(out, input_.contiguous(), group=self.device_group)
        out = out.reshape([self.world_size] + sizes0).movedim(0, dim)
        sizes = list(input_.size())
        sizes[dim] *= self.world_size
        return out.reshape(sizes)

    def broadcast(self, input_: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size > 1:
            dist.broadcast(input_, src=self.ranks[src], group=self.device_group)
        return input_

    def barrier(self) -> None:
        if self.world_size > 1:
            dist.barrier(group=self.device_group)


_TP: Optional[GroupCoordinator] = None


def init_distributed_environment(world_size: int = -1, rank: int = -1, local_rank: int = -1,
                                 distributed_init_method: str = "env://", backend: Optional[str] = None) -> None:
    """Initialises torch.distributed (if needed) and a TP group spanning all ranks (intra-node TP, as in BASELINE
    config 5: TP = 8 over the xGMI mesh)."""
    global _TP
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        world_size = world_size if world_size > 0 else int(os.environ.get("WORLD_SIZE", "1"))
        rank = rank if rank >= 0 else int(os.environ.get("RANK", "0"))
        dist.init_process_group(backend=backend, init_method=distributed_init_method, world_size=world_size, rank=rank)
    if local_rank < 0:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    _TP = GroupCoordinator(range(dist.get_world_size()), local_rank, backend)


def get_tp_group() -> GroupCoordinator:
    assert _TP is not None, "tensor model parallel group is not initialized"
    return _TP


def get_tensor_model_parallel_world_size() -> int:
    return 1 if _TP is None else _TP.world_size


def get_tensor_model_parallel_rank() -> int:
    return 0 if _TP is None else _TP.rank_in_group


def destroy_model_parallel() -> None:
    global _TP
    _TP = None
