"""GPU box: why does gptq_marlin_gemm_silu_and_mul differ from GEMM + silu_and_mul at (2, 14336, 896, g128)?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from neuralmagic_vllm_amd import _custom_ops as ops
from oracle import packing
import test_dispatch_fuzz_gpu as fz
M, N, K, group = 2, 14336, 896, 128
dtype = torch.float16 if (M + N // 64 + K // 128) % 3 else torch.bfloat16
a, packed, s, w_ref = fz._make(M, N, K, group, dtype, M + N + K)
e = torch.empty(0, dtype=torch.int32, device="cuda:0")
ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device="cuda:0")
mq = ops.gptq_marlin_repack(packed, e, K, N, 4)
ms = packing.marlin_permute_scales(s, K, N, group)
out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
out2 = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
two = torch.empty(M, N // 2, dtype=dtype, device="cuda:0")
ops.silu_and_mul(two, out)
one = ops.gptq_marlin_gemm_silu_and_mul(a, mq, ms, e, e, ws, 4, M, N, K, True)
one2 = ops.gptq_marlin_gemm_silu_and_mul(a, mq, ms, e, e, ws, 4, M, N, K, True)
torch.cuda.synchronize()
ref = a.float() @ w_ref.float()
print("lib", os.environ.get("NMX_LIB_PATH"), "dtype", dtype)
print("plain deterministic", torch.equal(out, out2), "fused deterministic", torch.equal(one, one2))
print("plain err", float((out.float() - ref).abs().mean() / ref.abs().mean()))
bad = (one.view(torch.int16) != two.view(torch.int16)).nonzero()
print("mismatches", bad.shape[0], "of", one.numel())
if bad.shape[0]:
    print("rows", bad[:, 0].unique().tolist(), "cols min/max", int(bad[:, 1].min()), int(bad[:, 1].max()))
    print("first", bad[:10].tolist())
    d = (one.float() - two.float()).abs()
    print("max abs diff", float(d.max()), "max rel", float((d / two.float().abs().clamp_min(1e-6)).max()))
    cols = bad[:, 1].unique()
    print("distinct cols", cols.numel(), cols[:40].tolist())
torch.save({"out": out.cpu(), "one": one.cpu()}, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "dbg_%s.pt" % ("r02" if os.environ.get("NMX_LIB_PATH") else "new")))
