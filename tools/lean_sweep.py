"""Times Marlin int4 GEMM configurations the way the decode step sees them: a chain of 32 launches over 32 distinct
weight tensors captured in a HIP graph (dependent-launch boundaries included, no profiler), and checks each forced
configuration against the default path of the skinny kernel on the same inputs.

usage (GPU box): python3 tools/lean_sweep.py [M ...] > gpurun_out/lean_sweep.txt
cfg strings: "D:auto" = default dispatch, "L:nw,splits[,mt[,ws]]" = marlin_decode_kernel, "G:ngrp" = marlin_large_kernel, "S:mt,ng,splits[,w8]" = marlin_gemm_kernel, "S:auto" its heuristic,
"X:splits" = marlin_dma_kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
if os.environ.get("LEAN_SWEEP_SET") == "kscan":  # gate_up's N at growing K: time = fixed cost of a launch + stages x cost of a stage
    SHAPES = {f"k{k}": (k, 28672) for k in (256, 512, 1024, 2048, 4096)}
if os.environ.get("LEAN_SWEEP_SET") == "70b-tp8":  # one rank of Llama-3-70B at TP = 8 (bench.py LLAMA3_70B_TP8_RANK)
    SHAPES = {"qkv": (8192, 1280), "o": (1024, 8192), "gate_up": (8192, 7168), "down": (3584, 8192)}
NL = 32
dev = "cuda:0"


def set_cfg(cfg):
    from neuralmagic_vllm_amd import _lib
    for k in ("NMX_GEMM_CFG", "NMX_GEMM_LEAN", "NMX_GEMM_LARGE", "NMX_GEMM_LARGE_NGRP", "NMX_GEMM_WIDE", "NMX_GEMM_DMA"):
        _lib.set_tuning(k, None)
    kind, val = cfg.split(":")
    if kind == "D":  # default dispatch, no override
        return
    if kind == "X":  # marlin_dma_kernel with val K splits
        _lib.set_tuning("NMX_GEMM_DMA", val)
        return
    _lib.set_tuning("NMX_GEMM_DMA", "0")
    if kind == "W":  # wide kernel "wm,wn,splits"
        _lib.set_tuning("NMX_GEMM_WIDE", val)
        return
    _lib.set_tuning("NMX_GEMM_WIDE", "0")
    if kind == "G":  # large-M kernel with 64 * val columns per workgroup
        _lib.set_tuning("NMX_GEMM_LARGE", "1")
        _lib.set_tuning("NMX_GEMM_LARGE_NGRP", val)
    elif kind == "L":
        _lib.set_tuning("NMX_GEMM_LEAN", val)
    else:
        _lib.set_tuning("NMX_GEMM_LEAN", "0")
        if val != "auto":
            _lib.set_tuning("NMX_GEMM_CFG", val)


def time_graph(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def cfgs_for(name, M):
    out = ["S:auto"]
    if M > 64:
        out = ["S:auto", "W:1,2,1", "W:1,2,2", "W:1,2,4", "W:1,4,1", "W:1,4,2", "W:1,4,4"]
        if M > 128:
            out += ["W:2,2,1", "W:2,2,2", "W:2,4,1", "W:2,4,2"]
        return out
    if M <= 16:
        out += ["L:16,1", "L:8,1", "L:8,2", "L:4,2", "L:4,3", "L:4,4", "L:8,4", "L:4,8", "L:8,1,1,0", "L:4,4,1,0", "S:1,1,2", "S:1,1,4", "S:1,1,8"]
        if name == "down":
            out += ["L:4,7", "L:4,14", "L:8,7", "S:1,1,7", "S:1,1,14"]
    elif M <= 32:
        out += ["L:8,1", "L:8,2", "L:4,2", "L:4,4", "L:8,1,2,0", "L:4,2,2,0", "S:2,2,2", "S:2,2,4"]
        if name == "down":
            out += ["L:4,7", "L:8,4"]
    else:
        out += ["S:2,2,2", "S:2,2,4", "S:4,4,2,1", "S:4,2,1,1", "S:4,2,2,1", "S:4,2,4,1"]
        if name == "down":
            out += ["L:4,7", "L:8,4", "S:2,2,8"]
    return out


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64]
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    wsp = torch.zeros(28672 // 64 * 16, dtype=torch.int32, device=dev)  # (>= N / 64 * 16 for every shape set)
    only_shapes = os.environ.get("LEAN_SWEEP_SHAPES")
    for name, (K, N) in SHAPES.items():
        if only_shapes and name not in only_shapes.split(","):
            continue
        ws = [(torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=dev, generator=g),
               (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()) for _ in range(NL)]
        for M in Ms:
            x = torch.randn(M, K, dtype=torch.float16, device=dev)
            set_cfg("S:auto")
            ref = ops.gptq_marlin_gemm(x, ws[0][0], ws[0][1], e, e, wsp, 4, M, N, K, True).float()
            torch.cuda.synchronize()
            by = K * N // 2 + (K // 128) * N * 2 + 2 * M * K + 2 * M * N
            only = os.environ.get("LEAN_SWEEP_ONLY")
            for cfg in (only.split(";") if only else cfgs_for(name, M)):
                if cfg.startswith("T:"):  # yardstick only: dense fp16 torch.matmul (hipBLASLt) on 32 distinct fp16 weights of the shape
                    dws = [torch.randn(K, N, dtype=torch.float16, device=dev) * 0.01 for _ in range(8 if K * N > 5e7 else NL)]

                    def run_t():
                        for i in range(NL):
                            torch.matmul(x, dws[i % len(dws)])
                    us = time_graph(run_t) / NL
                    print(f"{'hipBLASLt-fp16':18} {name:8} M={M:4d} {cfg:12} {us:7.2f} us  {(2 * K * N + 2 * M * K + 2 * M * N) / us / 1e3:7.0f} GB/s  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s  (dense fp16 yardstick)", flush=True)
                    del dws
                    continue
                set_cfg(cfg)
                try:
                    out = ops.gptq_marlin_gemm(x, ws[0][0], ws[0][1], e, e, wsp, 4, M, N, K, True).float()
                    torch.cuda.synchronize()
                    err = ((out - ref).abs().mean() / ref.abs().mean()).item()

                    defer = os.environ.get("LEAN_SWEEP_DEFER") == "1"  # GEMM only: the split-K slabs stay for a consumer

                    def run():
                        for w in ws:
                            if defer:
                                ops.gptq_marlin_gemm_deferred(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)
                            else:
                                ops.gptq_marlin_gemm(x, w[0], w[1], e, e, wsp, 4, M, N, K, True)

                    us = time_graph(run) / NL
                    print(f"{os.environ.get('NMX_LIB_PATH', '').split('/')[-1]:18} {name:8} M={M:4d} {cfg:12} {us:7.2f} us  {by / us / 1e3:7.0f} GB/s  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s  relerr_vs_default={err:.2e}", flush=True)
                except Exception as ex:  # noqa: BLE001
                    print(f"{name:8} M={M:3d} {cfg:12} FAILED {ex}", flush=True)
                    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
