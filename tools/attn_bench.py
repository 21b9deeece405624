"""Times paged_attention_v1 at the bench's decode shapes (Llama-3-8B heads, block 16) over 32 distinct KV caches in a
HIP graph. usage: python3 tools/attn_bench.py [batch ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops  # noqa: E402

dev = "cuda:0"
H, Hkv, D, BS, CTX, NL = 32, 8, 128, 16, 1024, 8
for B in [int(a) for a in sys.argv[1:]] or [64, 256]:
    nb = B * (CTX // BS)
    kvs = [(torch.empty(nb, Hkv, D // 8, BS, 8, dtype=torch.float16, device=dev).uniform_(-0.1, 0.1),
            torch.empty(nb, Hkv, D, BS, dtype=torch.float16, device=dev).uniform_(-0.1, 0.1)) for _ in range(NL)]
    bt = torch.randperm(nb, device=dev).to(torch.int32).reshape(B, -1)
    sl = torch.full((B, ), CTX, dtype=torch.int32, device=dev)
    q = torch.randn(B, H, D, dtype=torch.float16, device=dev) * 0.1
    out = torch.empty_like(q)

    def run():
        for kc, vc in kvs:
            ops.paged_attention_v1(out, q, kc, vc, Hkv, D**-0.5, bt, sl, BS, CTX, None, "auto", 1.0)

    run()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 / NL * 1e3
    by = 2 * B * CTX * Hkv * D * 2
    print(f"batch {B:4d}: {us:8.2f} us  {by / us / 1e3:7.0f} GB/s", flush=True)
