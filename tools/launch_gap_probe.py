"""How much does one dependent kernel launch cost inside a HIP graph on this GPU? Chains of tiny kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neuralmagic_vllm_amd import _custom_ops as ops

dev = "cuda:0"
x = torch.randn(1, 4096, dtype=torch.float16, device=dev)
w = torch.ones(4096, dtype=torch.float16, device=dev)
out = torch.empty_like(x)


def chain(n, kind):
    for _ in range(n):
        if kind == "rms":
            ops.rms_norm(out, x, w, 1e-5)
        elif kind == "torch_add":
            torch.add(x, 1.0, out=out)
        else:
            ops.silu_and_mul(out[:, :2048], x)


for kind in ("rms", "silu", "torch_add"):
    res = {}
    for n in (50, 200):
        chain(4, kind)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain(4, kind)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(n, kind)
        g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        res[n] = t0.elapsed_time(t1) / 10 * 1e3
    print(f"{kind:10}: {(res[200] - res[50]) / 150:.2f} us per dependent launch in a graph (50: {res[50]:.0f} us, 200: {res[200]:.0f} us)")
