"""CPU test (gloo, world_size 2) of the tensor-parallel path: Megatron column/row sharding of a GPTQ-Marlin MLP through
the weight-loader attributes + ONE sum all-reduce. There is no GPU here, so the per-rank GEMM is the CPU
oracle patched in at the `_custom_ops` level — the sharding, loader narrowing, repack bookkeeping and the collective are
the product's own code."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _patch_ops_with_oracle():
    """GEMM / repack of the Marlin family replaced by the CPU oracle (tests may use the oracle; the product may not)."""
    import oracle
    from oracle import packing
    from neuralmagic_vllm_amd import _custom_ops as ops

    def repack(b_q_weight, perm, size_k, size_n, num_bits):
        pf = 32 // num_bits
        q = torch.stack([(b_q_weight >> (num_bits * i)) & ((1 << num_bits) - 1) for i in range(pf)], dim=1).reshape(size_k, size_n)
        if perm.numel():
            q = q[perm.long()]
        return packing.marlin_weights(q, size_k, size_n, num_bits)

    ops.gptq_marlin_repack = repack
    ops.gptq_marlin_gemm = lambda a, q, s, g, p, ws, bits, m, n, k, full: oracle.gptq_marlin_gemm(a, q, s, g, p, ws, bits, m, n, k, full)
    ops.awq_gemm = oracle.awq_gemm
    ops.awq_dequantize = oracle.awq_dequantize


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neuralmagic_vllm_amd.distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                                                  init_distributed_environment, tensor_model_parallel_all_gather,
                                                  tensor_model_parallel_all_reduce)
    init_distributed_environment(backend="gloo")
    assert get_tensor_model_parallel_world_size() == world and get_tensor_model_parallel_rank() == rank
    # collectives
    t = torch.full((3, 4), float(rank + 1))
    assert torch.equal(tensor_model_parallel_all_reduce(t.clone()), torch.full((3, 4), float(sum(range(1, world + 1)))))
    g = tensor_model_parallel_all_gather(torch.full((2, 3), float(rank)), dim=-1)
    assert g.shape == (2, 3 * world) and torch.equal(g[:, 3 * rank:3 * rank + 3], torch.full((2, 3), float(rank)))

    _patch_ops_with_oracle()
    from oracle import packing
    from neuralmagic_vllm_amd.layers.linear import ColumnParallelLinear, RowParallelLinear
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig
    torch.manual_seed(0)  # same "checkpoint" on every rank
    H, I, M, G = 256, 512, 5, 128
    cfg = GPTQMarlinConfig(4, G, False, True)
    w_up, w_down = torch.randn(H, I, dtype=torch.float16) * 0.1, torch.randn(I, H, dtype=torch.float16) * 0.1
    x = torch.randn(M, H, dtype=torch.float16)
    ck = {}
    for name, w in (("up", w_up), ("down", w_down)):
        K, N = w.shape
        w_ref, q_w, s, _, _ = packing.quantize_weights(w, 4, G, False)
        ck[name] = dict(qweight=packing.gptq_pack(q_w, 4, K, N), scales=s, g_idx=torch.arange(K, dtype=torch.int32) // G, w_ref=w_ref)
    up = ColumnParallelLinear(H, I, cfg)
    down = RowParallelLinear(I, H, cfg)
    for layer, name in ((up, "up"), (down, "down")):
        for pname in ("qweight", "scales", "g_idx"):
            p = getattr(layer, pname)
            p.weight_loader(p, ck[name][pname])
    assert up.qweight.shape == (H // 8, I // world) and down.qweight.shape == (I // world // 8, H)
    assert down.scales.shape == (I // world // G, H)  # grouped scales are K-sharded on row-parallel layers
    h = up(x)                      # [M, I / world] — no communication
    y = down(h)                    # partial products summed by ONE all-reduce
    ref = (x.float() @ ck["up"]["w_ref"].float()).half().float() @ ck["down"]["w_ref"].float()
    err = float((y.float() - ref).abs().mean() / ref.abs().mean())
    ret[rank] = err
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_gptq_marlin_mlp_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert len(ret) == world
    for r in range(world):
        assert ret[r] < 2e-3, ret[r]
    assert abs(ret[0] - ret[1]) < 1e-9  # every rank holds the identical reduced result


def _worker_awq(rank, world, port, ret):
    """The AWQ shard rules of BASELINE config 5 (awq.py:76-176 under linear.py:408-445): column-parallel layers split the
    PACKED output dimension of qweight / qzeros (N / 8 words) and the scales on N, row-parallel layers split K - qweight
    rows, and qzeros / scales group rows - followed by the one sum all-reduce."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neuralmagic_vllm_amd.distributed import init_distributed_environment
    init_distributed_environment(backend="gloo")
    _patch_ops_with_oracle()
    from oracle import packing
    from neuralmagic_vllm_amd.layers.linear import MergedColumnParallelLinear, RowParallelLinear
    from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
    torch.manual_seed(0)
    H, I, M, G = 256, 512, 5, 128
    cfg = AWQConfig(4, G, True)
    gate_up = MergedColumnParallelLinear(H, [I, I], cfg)   # gate and up shards are loaded separately (llama.py:433-470)
    down = RowParallelLinear(I, H, cfg)
    w_gate, w_up, w_down = (torch.randn(H, I, dtype=torch.float16) * 0.1 for _ in range(2)), None, torch.randn(I, H, dtype=torch.float16) * 0.1
    w_gate, w_up = tuple(w_gate)
    refs = {}
    for name, w, layer, shard in (("gate", w_gate, gate_up, 0), ("up", w_up, gate_up, 1), ("down", w_down, down, None)):
        w_ref, qweight, qzeros, scales = packing.awq_quantize(w, G)
        refs[name] = w_ref
        for pname, tsr in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales)):
            prm = getattr(layer, pname)
            if shard is None:
                prm.weight_loader(prm, tsr)
            else:
                prm.weight_loader(prm, tsr, shard)
    assert gate_up.qweight.shape == (H, 2 * I // world // 8) and gate_up.scales.shape == (H // G, 2 * I // world)
    assert down.qweight.shape == (I // world, H // 8) and down.qzeros.shape == (I // world // G, H // 8)
    x = torch.randn(M, H, dtype=torch.float16)
    gu = gate_up(x)
    g, u = gu.chunk(2, dim=-1)
    y = down((g.float() * u.float()).half())
    gr, ur = (x.float() @ refs["gate"].float()).half().float(), (x.float() @ refs["up"].float()).half().float()
    ref = (gr * ur).half().float() @ refs["down"].float()
    ret[rank] = float((y.float() - ref).abs().mean() / ref.abs().mean())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_awq_mlp_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_awq, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for r in range(world):
        assert ret[r] < 3e-3, ret[r]
    assert abs(ret[0] - ret[1]) < 1e-9
