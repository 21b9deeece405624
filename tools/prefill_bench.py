"""Times context_attention_fwd (Llama-3-8B geometry) for a few (batch, new tokens, context) shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd

dev = "cuda:0"
from neuralmagic_vllm_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "0":
    _lib.set_tuning("NMX_PREFILL_GQ", sys.argv[1])
H, Hkv, D, BS = 32, 8, 128, 16
SHAPES = ((8, 1024, 0), (8, 512, 512), (4, 2048, 0), (64, 16, 1024), (1, 4096, 0))
if len(sys.argv) > 2:
    SHAPES = (SHAPES[int(sys.argv[2])], )
for batch, n_new, ctx in SHAPES:
    T = batch * n_new
    q = torch.randn(T, H, D, dtype=torch.float16, device=dev) * 0.1
    k = torch.randn(T, Hkv, D, dtype=torch.float16, device=dev) * 0.1
    v = torch.randn(T, Hkv, D, dtype=torch.float16, device=dev) * 0.1
    nblk = batch * ((ctx + BS - 1) // BS + 1)
    kc = torch.randn(nblk, Hkv, D // 8, BS, 8, dtype=torch.float16, device=dev) * 0.1
    vc = torch.randn(nblk, Hkv, D, BS, dtype=torch.float16, device=dev) * 0.1
    b_loc = torch.randperm(nblk, device=dev).to(torch.int32).reshape(batch, -1)
    start = (torch.arange(batch, device=dev) * n_new).to(torch.int32)
    sl = torch.full((batch, ), ctx + n_new, dtype=torch.int32, device=dev)
    cl = torch.full((batch, ), ctx, dtype=torch.int32, device=dev)
    o = torch.empty_like(q)
    for _ in range(2):
        context_attention_fwd(q, k, v, o, kc, vc, b_loc, start, sl, cl, n_new)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        context_attention_fwd(q, k, v, o, kc, vc, b_loc, start, sl, cl, n_new)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 5
    flops = 4.0 * batch * H * D * (n_new * ctx + n_new * (n_new + 1) / 2)
    print(f"batch {batch:3d} new {n_new:5d} ctx {ctx:5d}: {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TFLOP/s")
