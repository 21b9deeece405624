"""Runs ONE int4 Marlin GEMM configuration REPS times (for rocprofv3 --kernel-trace / --pmc passes).
usage: python3 tools/gemm_one.py SHAPE M "mt,ng,splits" [reps] [sparse24]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def main():
    name, M, cfg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    from neuralmagic_vllm_amd import _custom_ops as ops
    from neuralmagic_vllm_amd import _lib
    if cfg.startswith("X:"):
        _lib.set_tuning("NMX_GEMM_DMA", cfg[2:])
    elif cfg.startswith("W:"):
        _lib.set_tuning("NMX_GEMM_DMA", "0")
        _lib.set_tuning("NMX_GEMM_WIDE", cfg[2:])
    elif cfg != "auto":
        _lib.set_tuning("NMX_GEMM_WIDE", "0")
        _lib.set_tuning("NMX_GEMM_CFG", cfg)
    dev = "cuda:0"
    K, N = SHAPES[name]
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    ws = [(torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=dev, generator=g),
           (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()) for _ in range(4)]
    wsp = torch.zeros(N // 64 * 16, dtype=torch.int32, device=dev)
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    for r in range(reps):
        w = ws[r % 4]
        ops.gptq_marlin_gemm(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
