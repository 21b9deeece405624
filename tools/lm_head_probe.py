"""lm_head (plain fp16 library GEMM, outside the hot path) in the two weight layouts: x @ W.t() with W [vocab, hidden] (the
checkpoint's layout, what bench.py does) against x @ Wt with Wt [hidden, vocab] contiguous. usage: python3 tools/lm_head_probe.py"""
import torch

dev = "cuda:0"
V, H = 128256, 4096
w = torch.randn(V, H, device=dev, dtype=torch.float16) * 0.02
wt = w.t().contiguous()


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in (1, 16, 64, 256):
    x = torch.randn(M, H, device=dev, dtype=torch.float16)
    a = t(lambda: torch.matmul(x, w.t()))
    b = t(lambda: torch.matmul(x, wt))
    c = t(lambda: torch.nn.functional.linear(x, w))
    print(f"M={M:4d}  x @ W.t(): {a:7.1f} us ({V * H * 2 / a / 1e6:.2f} TB/s)   x @ Wt: {b:7.1f} us   F.linear: {c:7.1f} us", flush=True)
