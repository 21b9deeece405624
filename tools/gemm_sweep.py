"""Micro-benchmark of the Marlin-format GEMM under rocprofv3 --kernel-trace: per-config kernel durations.

usage (GPU box):
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/gemm_sweep.py run  LABELS.json
  python3 tools/gemm_sweep.py report OUT LABELS.json
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
REPS = 10


def configs():
    cfgs = []
    for M in (1, 16, 32, 64):
        for name in ("qkv", "o", "gate_up", "down"):
            if M <= 16:
                for mt, ng in ((1, 1), (1, 2), (1, 4)):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                        cfgs.append((name, M, mt, ng, sp))
            elif M <= 32:
                for mt, ng in ((2, 2), (2, 4)):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                        cfgs.append((name, M, mt, ng, sp))
            else:
                for mt, ng in ((4, 4), (2, 2), (2, 4)):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                        cfgs.append((name, M, mt, ng, sp))
    return cfgs


def run(labels_path):
    import torch
    from neuralmagic_vllm_amd import _custom_ops as ops
    from neuralmagic_vllm_amd import _lib
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    labels = []
    weights = {}
    for name, (K, N) in SHAPES.items():
        weights[name] = [(torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=dev, generator=g),
                          (torch.rand(K // 128, N, device=dev, generator=g) * 0.004 + 0.002).half()) for _ in range(3)]
    ws = torch.zeros(28672 // 64 * 16, dtype=torch.int32, device=dev)
    for (name, M, mt, ng, sp) in configs():
        K, N = SHAPES[name]
        x = torch.randn(M, K, dtype=torch.float16, device=dev)
        _lib.set_tuning("NMX_GEMM_CFG", f"{mt},{ng},{sp}")
        for r in range(REPS):
            w = weights[name][r % 3]
            ops.gptq_marlin_gemm(x, w[0], w[1], e, e, ws, 4, M, N, K, True)
        torch.cuda.synchronize()
        labels.append(dict(name=name, M=M, mt=mt, ng=ng, splits=sp, launches=REPS, reduce=sp > 1))
    json.dump(labels, open(labels_path, "w"))


def report(out_dir, labels_path):
    import csv
    files = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gemm = [r for r in rows if "marlin_gemm_kernel" in r["Kernel_Name"]]
    red = [r for r in rows if "splitk_reduce" in r["Kernel_Name"]]
    labels = json.load(open(labels_path))
    gi = ri = 0
    print(f"{'shape':8} {'M':>3} {'mt':>2} {'ng':>2} {'sp':>2} {'gemm_us':>8} {'red_us':>7} {'GB/s(gemm+red)':>14} {'vgpr':>5} {'lds':>6}")
    for lb in labels:
        n = lb["launches"]
        gs = gemm[gi:gi + n]
        gi += n
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in gs)[n // 2] / 1e3
        rd = 0.0
        if lb["reduce"]:
            rs = red[ri:ri + n]
            ri += n
            rd = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)[n // 2] / 1e3
        K, N = SHAPES[lb["name"]]
        by = K * N // 2 + (K // 128) * N * 2 + 2 * lb["M"] * (K + N)
        print(f"{lb['name']:8} {lb['M']:3d} {lb['mt']:2d} {lb['ng']:2d} {lb['splits']:2d} {d:8.2f} {rd:7.2f} {by / (d + rd) / 1e3:14.1f} "
              f"{gs[0].get('VGPR_Count', gs[0].get('Arch_VGPR_Count', '?')):>5} {gs[0].get('LDS_Block_Size', '?'):>6}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3])
