"""Runs ONE cutlass_scaled_mm shape REPS times (for rocprofv3 kernel traces): python3 tools/mm_one.py M N K [fp8|int8] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neuralmagic_vllm_amd import _custom_ops as ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fp8 = (sys.argv[4] if len(sys.argv) > 4 else "fp8") == "fp8"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
if fp8:
    a = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
    bs = [torch.randn(N, K, device=dev).to(torch.float8_e4m3fn) for _ in range(4)]
else:
    a = torch.randint(-100, 100, (M, K), device=dev, dtype=torch.int8)
    bs = [torch.randint(-100, 100, (N, K), device=dev, dtype=torch.int8) for _ in range(4)]
sa = torch.ones(1, device=dev)
sb = torch.ones(1, device=dev)
for r in range(reps):
    ops.cutlass_scaled_mm(a, bs[r % 4].t(), sa, sb, torch.float16)
torch.cuda.synchronize()
