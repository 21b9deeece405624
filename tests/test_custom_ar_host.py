"""CPU tests of the one-shot xGMI all-reduce's host logic (no GPU): the dispatch thresholds of the reference
(csrc/custom_all_reduce.cuh:442-450, custom_all_reduce.cu should_custom_ar), the rank-ordered IPC-meta exchange over a
gloo group (custom_all_reduce.py:188-216) and the gates."""
import ctypes
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_thresholds():
    from neuralmagic_vllm_amd.distributed.custom_all_reduce import custom_ar_scratch_bytes, custom_ar_stages, should_custom_ar
    mx = 8192 * 1024
    assert should_custom_ar(4 << 20, mx, 2, True) and should_custom_ar(4 << 20, mx, 2, False)   # two ranks: any size up to max
    assert not should_custom_ar(mx + 16, mx, 2, True)
    for w in (4, 6, 8):
        assert should_custom_ar(mx, mx, w, True) and not should_custom_ar(mx + 16, mx, w, True)
    assert not should_custom_ar(1024, mx, 4, False)       # more than two ranks need the full mesh
    assert not should_custom_ar(1000, mx, 2, True)        # 16-byte multiples only
    # schedule choice of custom_all_reduce.cuh:442-450: two ranks always one-stage; <= 4 ranks below 512 KiB, <= 8 below 256 KiB
    assert custom_ar_stages(4 << 20, 2) == 1
    assert custom_ar_stages(512 * 1024 - 16, 4) == 1 and custom_ar_stages(512 * 1024, 4) == 2
    for w in (6, 8):
        assert custom_ar_stages(256 * 1024 - 16, w) == 1 and custom_ar_stages(256 * 1024, w) == 2
    # the decode message of BASELINE configs[4]: batch 8 = [8, 8192] fp16 = 128 KiB -> one-stage; batch 256 = 4 MiB -> two-stage
    assert custom_ar_stages(8 * 8192 * 2, 8) == 1 and custom_ar_stages(256 * 8192 * 2, 8) == 2
    # two-stage scratch per rank: its slice of the 16-byte packets, the last rank also takes the remainder
    assert custom_ar_scratch_bytes(256 * 8192 * 2, 8) == 256 * 8192 * 2 // 8
    assert custom_ar_scratch_bytes(16 * 21, 4) == (21 // 4 + 21 % 4) * 16
    assert custom_ar_scratch_bytes(mx, 6) >= mx // 6


def test_disabled_without_gate(monkeypatch):
    from neuralmagic_vllm_amd.distributed.custom_all_reduce import CustomAllreduce
    monkeypatch.delenv("NMX_CUSTOM_AR", raising=False)
    ca = CustomAllreduce(group=None, device="cuda:0")
    assert ca.disabled and ca.custom_all_reduce(torch.zeros(8)) is None


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from neuralmagic_vllm_amd.distributed.custom_all_reduce import gather_ipc_meta
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group(list(range(world)), backend="gloo")
    shard = (bytes([rank + 1]) * 64, 4096 * (rank + 1))  # what _share_cuda_() would report: (64-byte handle, offset)
    handles, offsets = gather_ipc_meta(group, rank, world, shard)
    ret[rank] = (handles, offsets)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ipc_meta_exchange_is_rank_ordered():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for r in range(world):
        handles, offsets = ret[r]
        assert handles == [bytes([1]) * 64, bytes([2]) * 64] and offsets == [4096, 8192]
