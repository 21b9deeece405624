"""Times gptq_marlin_24_gemm (2:4-sparse int4, group 128) configurations the way the decode step sees them: 32-launch HIP-graph
chains over 32 distinct weight tensors, deferred reduce (the step's consumers sum the K-split slabs).

usage (GPU box): python3 tools/sparse_sweep.py "128 256" "D;0;1,4,1;1,2,2" [shapes]   (cfg = NMX_GEMM_WIDE value, D = default)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402
from tools.lean_sweep import time_graph  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
NL = 32
dev = "cuda:0"


def valid_meta(rows, cols, gen):
    nib = torch.tensor([0x4, 0x8, 0xC, 0x9, 0xD, 0xE], dtype=torch.int32, device=dev)
    pick = nib[torch.randint(0, 6, (rows, cols, 4), device=dev, generator=gen)]
    v = pick[..., 0] | (pick[..., 1] << 4) | (pick[..., 2] << 8) | (pick[..., 3] << 12)
    return v.to(torch.int16)


def main():
    Ms = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "256").split()]
    cfgs = (sys.argv[2] if len(sys.argv) > 2 else "D").split(";")
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(SHAPES)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    for name in names:
        K, N = SHAPES[name]
        ws = [(torch.randint(-2**31, 2**31 - 1, (K // 32, N * 2), dtype=torch.int32, device=dev, generator=g), valid_meta(K // 32, N * 2, g),
               (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()) for _ in range(NL)]
        wsp = torch.zeros(N // 128 * 64, dtype=torch.int32, device=dev)
        for M in Ms:
            x = torch.randn(M, K, dtype=torch.float16, device=dev)
            _lib.set_tuning("NMX_GEMM_WIDE", "0")
            ref = ops.gptq_marlin_24_gemm(x, ws[0][0], ws[0][1], ws[0][2], wsp, 4, M, N, K).float()
            for cfg in cfgs:
                _lib.set_tuning("NMX_GEMM_WIDE", None if cfg == "D" else cfg)
                out = ops.gptq_marlin_24_gemm(x, ws[0][0], ws[0][1], ws[0][2], wsp, 4, M, N, K).float()
                torch.cuda.synchronize()
                err = ((out - ref).abs().mean() / ref.abs().mean()).item()

                def run():
                    for w in ws:
                        ops.gptq_marlin_24_gemm_deferred(x, w[0], w[1], w[2], wsp, 4, M, N, K)

                us = time_graph(run) / NL
                by = K * N // 4 + K * N // 8 + (K // 128) * N * 2 + 2 * M * K + 2 * M * N
                print(f"{name:8} M={M:4d} {cfg:10} {us:7.2f} us  {by / us / 1e3:7.0f} GB/s  {2.0 * M * K * N / us / 1e6:7.1f} dense-equivalent TFLOP/s  "
                      f"relerr_vs_row_block_kernel={err:.2e}", flush=True)


if __name__ == "__main__":
    main()
