"""Times ONE int4 Marlin GEMM configuration with HIP events: python3 tools/gemm_time.py SHAPE M CFG [reps]
CFG = "auto" | "W:wm,wn,splits" | "mt,ng,splits[,w8]". Weights rotate over 8 tensors (no L2 / MALL reuse between calls)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def main():
    name, M, cfg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    nt = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    from neuralmagic_vllm_amd import _custom_ops as ops
    from neuralmagic_vllm_amd import _lib
    if cfg.startswith("X:"):
        _lib.set_tuning("NMX_GEMM_DMA", cfg[2:])
    elif cfg.startswith("W:"):
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
           (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()) for _ in range(nt)]
    wsp = torch.zeros(N // 64 * 16, dtype=torch.int32, device=dev)
    x = torch.randn(M, K, dtype=torch.float16, device=dev)
    for w in ws:
        ops.gptq_marlin_gemm(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for r in range(reps):
            w = ws[r % nt]
            ops.gptq_marlin_gemm(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / reps * 1e3)
    print(f"{os.environ.get('NMX_LIB_PATH', 'default').split('/')[-1]:28} {name:8} M={M:4d} {cfg:10} {best:8.2f} us  {2.0 * M * K * N / best / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
