"""GPU sanity test of the tensor-parallel path over RCCL (torch.distributed backend "nccl") with the real HIP GEMM.
One GPU is available to the test, so world_size is 1: it exercises process-group creation on the device, the
all-reduce / all-gather calls on device tensors and a Column -> Row parallel GPTQ-Marlin MLP end to end. The sharding
arithmetic for world_size 2 is covered on CPU by tests/test_tp_gloo.py."""
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
