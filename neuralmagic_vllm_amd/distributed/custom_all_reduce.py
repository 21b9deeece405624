"""All-reduce over the xGMI mesh - host side. Mirrors vllm/distributed/device_communicators/custom_all_reduce.py
(CustomAllreduce: gates :37-131, IPC meta exchange :171-216, dispatch :218-262) on top of the nmx_custom_ar_* C-ABI.

Differences from the reference, all deliberate:
  * both schedules of the reference exist (one-stage below 512 KiB at <= 4 ranks / 256 KiB at 6-8 ranks, two-stage above,
    csrc/custom_all_reduce.cuh:442-450), each rank reading every peer over its own xGMI link; messages beyond `max_size`
    return None and the caller falls back to RCCL;
  * the registered staging buffer is used in every mode (eager and graph capture): the copy into it is a capturable
    device-to-device copy, so no graph-buffer registration pass is needed;
  * the signal + scratch block is uncached fine-grained device memory from nmx_custom_ar_alloc_meta (flags polled while
    peers write them over xGMI), not a cached torch allocation;
  * a barrier that times out sets an error word and that call writes no sum: `check()` reads it (after a sync; call it after
    a graph replay), NMX_CUSTOM_AR_CHECK=1 makes every eager call sync and raise;
  * OFF unless NMX_CUSTOM_AR=1: no multi-GPU node has measured it yet (the one-GPU loop-back test proves the kernels, the
    epoch barriers and the cross-XCD visibility of flags and scratch, not the IPC mappings).
"""
import ctypes
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from neuralmagic_vllm_amd import _lib

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def should_custom_ar(nbytes: int, max_size: int, world_size: int, full_xgmi: bool) -> bool:
    return bool(_lib.lib().nmx_custom_ar_should(ctypes.c_int64(nbytes), ctypes.c_int64(max_size), ctypes.c_int(world_size),
                                                ctypes.c_int(int(full_xgmi))))


def custom_ar_stages(nbytes: int, world_size: int) -> int:
    """1 = one-stage (every rank reads every peer), 2 = two-stage (reduce-scatter + all-gather): csrc/custom_all_reduce.cuh:442-450."""
    return int(_lib.lib().nmx_custom_ar_stages(ctypes.c_int64(nbytes), ctypes.c_int(world_size)))


def custom_ar_scratch_bytes(nbytes: int, world_size: int) -> int:
    """Peer-visible scratch a rank needs behind its signal block for a two-stage call on `nbytes` (its slice + the remainder)."""
    lib = _lib.lib()
    lib.nmx_custom_ar_scratch_bytes.restype = ctypes.c_int64
    return int(lib.nmx_custom_ar_scratch_bytes(ctypes.c_int64(nbytes), ctypes.c_int(world_size)))


def gather_ipc_meta(group, rank: int, world_size: int, shard: Tuple[bytes, int]) -> Tuple[List[bytes], List[int]]:
    """Every rank's (handle, offset) in RANK ORDER (custom_all_reduce.py:188-216: one broadcast per rank; all_gather_object
    is avoided for the same gloo / inference-mode reason as there)."""
    all_data = [[None] for _ in range(world_size)]
    all_data[rank][0] = shard
    ranks = sorted(dist.get_process_group_ranks(group=group))
    for i, r in enumerate(ranks):
        dist.broadcast_object_list(all_data[i], src=r, group=group, device="cpu")
    return [d[0][0] for d in all_data], [d[0][1] for d in all_data]


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group, device, max_size: int = 8192 * 1024) -> None:
        self.disabled = True
        self._ptr = None
        self._opened = []
        if os.environ.get("NMX_CUSTOM_AR", "0") != "1":
            return  # gate: not measured on a multi-GPU node yet
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, "CustomAllreduce should be attached to a non-NCCL group."
        rank, world_size = dist.get_rank(group=group), dist.get_world_size(group=group)
        if world_size == 1 or world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.rank, self.world_size, self.max_size = rank, world_size, max_size
        # one MI355X node is a full xGMI mesh; ranks of one node only (the caller guarantees a single-node group)
        self.full_xgmi = True
        lib = _lib.lib()
        lib.nmx_custom_ar_meta_size.restype = ctypes.c_int64
        lib.nmx_custom_ar_scratch_bytes.restype = ctypes.c_int64
        self.scratch_bytes = int(lib.nmx_custom_ar_scratch_bytes(ctypes.c_int64(max_size), ctypes.c_int(world_size)))
        meta_bytes = int(lib.nmx_custom_ar_meta_size()) + self.scratch_bytes
        meta = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.nmx_custom_ar_alloc_meta(ctypes.c_int64(meta_bytes), ctypes.byref(meta)))
        self._meta = meta
        self.buffer = torch.empty(max_size, dtype=torch.uint8, device=self.device)
        handle = ctypes.create_string_buffer(64)
        _lib.check(lib.nmx_ipc_get_mem_handle(meta, handle))
        sig_ptrs = self._exchange_raw(meta.value, (bytes(handle.raw), 0))
        fa = ctypes.c_void_p()
        _lib.check(lib.nmx_custom_ar_init((ctypes.c_void_p * world_size)(*sig_ptrs), ctypes.c_int(rank), ctypes.c_int(world_size),
                                          ctypes.c_int64(self.scratch_bytes), ctypes.byref(fa)))
        self._ptr = fa
        self.register_buffer(self.buffer)
        self.check_every_call = os.environ.get("NMX_CUSTOM_AR_CHECK", "0") == "1"
        self.disabled = False

    # -- IPC -----------------------------------------------------------------------------------------------------
    def _ipc_meta(self, t: torch.Tensor) -> Tuple[bytes, int]:
        """(handle of the allocation that holds t, byte offset of t inside it) - as _share_cuda_() reports them
        (custom_all_reduce.py:171-178)."""
        data = t.untyped_storage()._share_cuda_()
        return bytes(data[1]), int(data[3])

    def _exchange(self, t: torch.Tensor) -> List[int]:
        """Pointers to every rank's copy of `t` as mapped in this process (own pointer for the own rank)."""
        return self._exchange_raw(t.data_ptr(), self._ipc_meta(t))

    def _exchange_raw(self, own_ptr: int, shard: Tuple[bytes, int]) -> List[int]:
        handles, offsets = gather_ipc_meta(self.group, self.rank, self.world_size, shard)
        ptrs = []
        for r in range(self.world_size):
            if r == self.rank:
                ptrs.append(own_ptr)
                continue
            base = ctypes.c_void_p()
            _lib.check(_lib.lib().nmx_ipc_open_mem_handle(ctypes.c_char_p(handles[r]), ctypes.byref(base)))
            self._opened.append(base)
            ptrs.append(base.value + offsets[r])
        return ptrs

    def register_buffer(self, inp: torch.Tensor) -> None:
        ptrs = self._exchange(inp)
        _lib.check(_lib.lib().nmx_custom_ar_register_buffer(self._ptr, (ctypes.c_void_p * self.world_size)(*ptrs)))

    # -- dispatch --------------------------------------------------------------------------------------------------
    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        return inp.dtype in _DT and inp.is_contiguous() and should_custom_ar(inp.numel() * inp.element_size(), self.max_size,
                                                                             self.world_size, self.full_xgmi)

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """Out-of-place sum over the group, or None when this communicator does not take the message (the caller then
        uses RCCL) - the contract of custom_all_reduce.py:238-262."""
        if self.disabled or not self.should_custom_ar(input):
            return None
        nbytes = input.numel() * input.element_size()
        staged = self.buffer[:nbytes].view(input.dtype).view(input.shape)
        staged.copy_(input)  # capturable D2D copy into the registered buffer
        out = torch.empty_like(input)
        stream = ctypes.c_void_p(torch.cuda.current_stream(input.device).cuda_stream)
        _lib.check(_lib.lib().nmx_custom_ar_all_reduce(self._ptr, ctypes.c_void_p(self.buffer.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                       ctypes.c_int64(input.numel()), ctypes.c_int(_DT[input.dtype]), stream))
        if self.check_every_call and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(input.device).synchronize()
            self.check()
        return out

    def check(self) -> None:
        """Raises if a barrier of any call since the last check timed out (a late or missing peer: that call wrote no sum).
        Reads the error word of this rank's signal block: call it after the stream / a graph replay has been synchronised."""
        if self._ptr is None:
            return
        err = ctypes.c_int(0)
        _lib.check(_lib.lib().nmx_custom_ar_check(self._ptr, ctypes.c_int(1), ctypes.byref(err)))
        if err.value != 0:
            raise RuntimeError("nmx custom all-reduce: a mesh barrier timed out (peer missing or not launched); the affected "
                               "call's output is not a sum")

    def close(self) -> None:
        if self._ptr:
            _lib.lib().nmx_custom_ar_dispose(self._ptr)
            self._ptr = None
        for base in self._opened:
            _lib.lib().nmx_ipc_close_mem_handle(base)
        self._opened = []
        if getattr(self, "_meta", None):
            _lib.lib().nmx_custom_ar_free_meta(self._meta)
            self._meta = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass
