"""Times fp8 cutlass_scaled_mm tile configurations the way the decode step sees them: a chain of 32 launches over 32 distinct
weight tensors in a HIP graph (no profiler), each forced configuration checked against the default path on the same inputs.

usage (GPU box): python3 tools/mm_sweep.py "256 512" "D;4,0,0;4,0,1;2,0,1" [shapes] > gpurun_out/mm_sweep.txt
cfg = NMX_MM_TILE value ("wn,splits,form"; D = default dispatch). MM_SWEEP_DEFER=1 leaves the split-K slabs to a consumer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
NL = 32
dev = "cuda:0"


def time_graph(fn, reps=5):
    fn()
    torch.cuda.synchronize()
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


def main():
    Ms = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "256").split()]
    cfgs = (sys.argv[2] if len(sys.argv) > 2 else "D").split(";")
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(SHAPES)
    defer = os.environ.get("MM_SWEEP_DEFER") == "1"
    sa = torch.full((1,), 0.02, device=dev)
    sb = torch.full((1,), 0.01, device=dev)
    for name in names:
        K, N = SHAPES[name]
        ws = [torch.randn(N, K, device=dev).to(torch.float8_e4m3fn) for _ in range(NL)]
        for M in Ms:
            a = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
            _lib.set_tuning("NMX_MM_TILE", None)
            ref = ops.cutlass_scaled_mm(a, ws[0].t(), sa, sb, torch.float16).float()
            exact = (a.float() @ ws[0].float().t()) * (0.02 * 0.01)
            for cfg in cfgs:
                if cfg == "T":  # yardstick only: torch._scaled_mm (hipBLASLt), never part of the product path
                    try:
                        def run_t():
                            for w in ws:
                                torch._scaled_mm(a, w.t(), scale_a=sa, scale_b=sb, out_dtype=torch.float16)
                        us = time_graph(run_t) / NL
                        print(f"{name:8} M={M:4d} {'hipBLASLt':8} {us:7.2f} us  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s  (torch._scaled_mm yardstick)", flush=True)
                    except Exception as ex:  # noqa: BLE001
                        print(f"{name:8} M={M:4d} hipBLASLt FAILED {ex}", flush=True)
                        torch.cuda.synchronize()
                    continue
                _lib.set_tuning("NMX_MM_TILE", None if cfg == "D" else cfg)
                try:
                    out = ops.cutlass_scaled_mm(a, ws[0].t(), sa, sb, torch.float16).float()
                    torch.cuda.synchronize()
                    err = ((out - exact).abs().mean() / exact.abs().mean()).item()
                    same = bool(torch.equal(out, ref))

                    def run():
                        for w in ws:
                            if defer:
                                ops.cutlass_scaled_mm_deferred(a, w.t(), sa, sb, torch.float16)
                            else:
                                ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.float16)

                    us = time_graph(run) / NL
                    by = K * N + M * K + 2 * M * N
                    print(f"{name:8} M={M:4d} {cfg:8} {us:7.2f} us  {by / us / 1e3:7.0f} GB/s  {2.0 * M * K * N / us / 1e6:7.1f} TFLOP/s  "
                          f"relerr_vs_fp32={err:.2e} same_bits_as_default={same}", flush=True)
                except Exception as ex:  # noqa: BLE001
                    print(f"{name:8} M={M:4d} {cfg:8} FAILED {ex}", flush=True)
                    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
