"""Checks kernel configurations forced through NMX_GEMM_CFG against the default configuration (itself covered by the
parity tests) on random int4 g128 weights, and prints event-timed durations. usage: python3 tools/check_cfg.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops  # noqa: E402
from neuralmagic_vllm_amd import _lib  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
CFGS = sys.argv[1].split(";") if len(sys.argv) > 1 else ["4,4,2,1", "4,4,1,1", "4,4,4,1", "4,4,2", "4,4,4"]
dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(0)
e = torch.empty(0, dtype=torch.int32, device=dev)
for name, (K, N) in SHAPES.items():
    w = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=dev, generator=g)
    s = (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()
    wsp = torch.zeros(N // 64 * 16, dtype=torch.int32, device=dev)
    for M in (64, 100):
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        _lib.set_tuning("NMX_GEMM_CFG", None)
        ref = ops.gptq_marlin_gemm(x, w, s, e, e, wsp, 4, M, N, K, True).float()
        for cfg in CFGS:
            _lib.set_tuning("NMX_GEMM_CFG", cfg)
            out = ops.gptq_marlin_gemm(x, w, s, e, e, wsp, 4, M, N, K, True).float()
            err = float((out - ref).abs().mean() / ref.abs().mean())
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                ops.gptq_marlin_gemm(x, w, s, e, e, wsp, 4, M, N, K, True)
            t1.record()
            torch.cuda.synchronize()
            print(f"{name:8} M={M:4d} cfg={cfg:10} rel_err_vs_default={err:.2e} {t0.elapsed_time(t1) / 20 * 1e3:8.1f} us/call"
                  f"{'  MISMATCH' if err > 2e-3 else ''}")
