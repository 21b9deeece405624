import sys, torch
sys.path.insert(0, '.')
from neuralmagic_vllm_amd import _custom_ops as ops
import oracle
codes = torch.arange(256, dtype=torch.int32).to(torch.uint8)
for kv in ("fp8", "fp8_e5m2"):
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        for scale in (1.0, 0.5):
            o = torch.empty(256, dtype=dt); oracle.convert_fp8(o, codes, scale, kv)
            g = torch.empty(256, dtype=dt, device="cuda"); ops.convert_fp8(g, codes.cuda(), scale, kv)
            g = g.cpu()
            bad = [(int(c), float(o[c]), float(g[c])) for c in range(256) if not (torch.isnan(o[c]) and torch.isnan(g[c])) and o[c].view(torch.int16 if dt != torch.float32 else torch.int32) != g[c].view(torch.int16 if dt != torch.float32 else torch.int32)]
            print(kv, dt, scale, "mismatches:", bad[:10])
