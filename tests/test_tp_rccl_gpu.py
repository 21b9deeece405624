"""GPU sanity test of the tensor-parallel path over RCCL (torch.distributed backend "nccl") with the real HIP GEMM.
One GPU is available to the builder's test box, so the first tests run at world_size 1: process-group creation on the device,
the all-reduce / all-gather calls on device tensors and a Column -> Row parallel GPTQ-Marlin MLP end to end. The two-rank legs
(RCCL all-reduce eager + captured in a HIP graph, the one-shot xGMI all-reduce over IPC) run wherever two GPUs are visible
(the driver's node) and FAIL there if the sum is wrong - they are skipped only for lack of a second GPU. The sharding arithmetic
for world_size 2 is covered on CPU by tests/test_tp_gloo.py."""
import os
import socket

import pytest
import torch

from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_tp_world1_rccl_mlp(ops):
    import torch.distributed as dist
    from neuralmagic_vllm_amd.distributed import (destroy_model_parallel, get_tensor_model_parallel_world_size,
                                                  init_distributed_environment, tensor_model_parallel_all_gather,
                                                  tensor_model_parallel_all_reduce)
    from neuralmagic_vllm_amd.layers.linear import ColumnParallelLinear, RowParallelLinear
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    init_distributed_environment(backend="nccl")
    try:
        assert get_tensor_model_parallel_world_size() == 1
        t = torch.arange(12, dtype=torch.float16, device=DEV).reshape(3, 4)
        assert torch.equal(tensor_model_parallel_all_reduce(t.clone()), t)
        assert torch.equal(tensor_model_parallel_all_gather(t, dim=-1), t)
        # the product bypasses the collective at world size 1 like the reference (parallel_state.py:273-276); call RCCL
        # itself on device tensors here so that the communicator, the kernel launch and graph capture are exercised
        from neuralmagic_vllm_amd.distributed import get_tp_group
        grp = get_tp_group().device_group
        u = t.clone()
        dist.all_reduce(u, group=grp)
        gat = torch.empty(1, 3, 4, dtype=torch.float16, device=DEV)
        dist.all_gather_into_tensor(gat, t, group=grp)
        torch.cuda.synchronize()
        assert torch.equal(u, t) and torch.equal(gat[0], t)
        # (no graph-capture claim at world size 1: RCCL elides a one-rank all-reduce, the captured graph is EMPTY and a
        # replay proves nothing - the capture of the collective is tested where two GPUs exist, test_tp2_* below)
        seed_all(0)
        H, I, G = 256, 512, 128
        cfg = GPTQMarlinConfig(4, G, False, True)
        up, down = ColumnParallelLinear(H, I, cfg), RowParallelLinear(I, H, cfg)
        refs = {}
        for layer, name, (K, N) in ((up, "up", (H, I)), (down, "down", (I, H))):
            w = torch.randn(K, N, dtype=torch.float16) * 0.1
            w_ref, q_w, sc, _, _ = packing.quantize_weights(w, 4, G, False)
            ck = dict(qweight=packing.gptq_pack(q_w, 4, K, N), scales=sc, g_idx=torch.arange(K, dtype=torch.int32) // G)
            for pname, tsr in ck.items():
                prm = getattr(layer, pname)
                prm.weight_loader(prm, tsr)
            for _, prm in layer.named_parameters():
                if not prm.is_meta:
                    prm.data = prm.data.to(DEV)
            refs[name] = w_ref
        x = torch.randn(5, H, dtype=torch.float16)
        y = down(up(x.to(DEV)))
        ref = (x.float() @ refs["up"].float()).half().float() @ refs["down"].float()
        assert compute_max_diff(y.cpu(), ref) < 2e-3
    finally:
        destroy_model_parallel()
        if dist.is_initialized():
            dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_tp_mode_one_rank():
    """bench.py --tp: the sharded step with the RCCL all-reduce inside the captured graph (world of one here; the driver
    runs the 2 / 4 / 8-rank legs). The JSON line must carry tp and the per-all-reduce figures."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "awq70b", "--tp", "1", "--gpus", "1",
                          "--layers", "2", "--batch", "8", "--ctx", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["tp"] == 1 and r["n_gpus"] == 1 and r["scaling"] == "strong" and r["config"]["hip_graph"] is True
    assert r["allreduce"]["per_step"] == 4 and r["allreduce"]["bytes"] == 8 * 8192 * 2 and r["allreduce"]["us"] > 0
    assert r["rccl"]["nranks"] == 1 and r["rccl"]["backend"] == "nccl"


def _tp2_worker(rank, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from neuralmagic_vllm_amd.distributed import init_distributed_environment, tensor_model_parallel_all_reduce
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    init_distributed_environment(backend="nccl")
    res = {}
    t = torch.full((256, 8192), float(rank + 1), dtype=torch.float16, device=dev)
    tensor_model_parallel_all_reduce(t)
    torch.cuda.synchronize()
    res["eager_mean"] = float(t.float().mean())
    # the same call CAPTURED in a HIP graph on the compute stream (bench.py --tp, the reference's pynccl path
    # device_communicators/pynccl.py:99-118): the replay must redo the sum on new contents of the buffer
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        v = torch.zeros(64, 8192, dtype=torch.float16, device=dev)
        tensor_model_parallel_all_reduce(v)  # communicator warm-up outside the capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            w = tensor_model_parallel_all_reduce(v)
        sums = []
        for it in range(3):
            v.fill_(float((rank + 1) * (it + 1)))
            g.replay()
            side.synchronize()
            sums.append(float(w.float().mean()))
    res["graph_sums"] = sums
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_tp2_rccl_all_reduce_two_gpus():
    """Two ranks, one GPU each: the [M, hidden] fp16 sum all-reduce of RowParallelLinear over RCCL / xGMI, eager AND captured
    in a HIP graph (the replayed graph must carry the collective: a sum of fresh buffer contents on every replay)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_tp2_worker, args=(r, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for r in range(2):
        assert ret[r]["eager_mean"] == 3.0
        assert ret[r]["graph_sums"] == [3.0, 6.0, 9.0], ret[r]


def _ca2_worker(rank, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", NMX_CUSTOM_AR="1")
    import torch.distributed as dist
    from neuralmagic_vllm_amd.distributed.custom_all_reduce import CustomAllreduce
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    group = dist.new_group([0, 1], backend="gloo")
    ca = CustomAllreduce(group, dev, max_size=1 << 20)
    res = {"disabled": ca.disabled}
    if not ca.disabled:
        outs = []
        for it, numel in enumerate((4096, 8 * 8192, 64 * 8192)):
            g = torch.Generator(device=dev)
            g.manual_seed(100 * it + rank)
            x = torch.randn(numel, device=dev, generator=g).to(torch.float16)
            y = ca.custom_all_reduce(x)
            torch.cuda.synchronize()
            ca.check()
            # expected: both ranks' inputs summed in rank order in fp32
            xs = []
            for r in range(2):
                gg = torch.Generator(device=dev)
                gg.manual_seed(100 * it + r)
                xs.append(torch.randn(numel, device=dev, generator=gg).to(torch.float16).float())
            outs.append(bool(torch.equal(y, (xs[0] + xs[1]).to(torch.float16))))
            dist.barrier()
        res["ok"] = outs
    ret[rank] = res
    ca.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_custom_all_reduce_two_gpus_ipc():
    """The one-shot xGMI all-reduce across a real device boundary: IPC handles exchanged over gloo, peer reads over xGMI, the
    fixed-order fp32 sum identical on both ranks, no barrier timeout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_ca2_worker, args=(r, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for r in range(2):
        assert ret[r]["disabled"] is False and ret[r]["ok"] == [True, True, True], ret[r]
