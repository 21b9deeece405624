"""Times paged_attention_v2 (and v1) over (batch, context) with forced partition sizes, the way the decode step sees the op:
a chain over 8 distinct KV caches in a HIP graph. Checks the rule of nmx_paged_attention_partition_size() beyond the bench's
1,024-token context. usage (GPU box): python3 tools/attn_part_sweep.py "1 4 8 16" "1024 4096 8192" > gpurun_out/attn_part_sweep.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops, _lib  # noqa: E402

dev = "cuda:0"
H, Hkv, D, BS, NL = 32, 8, 128, 16, 8


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
    return e0.elapsed_time(e1) / reps * 1e3 / NL


def main():
    batches = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "1 4 8 16").split()]
    ctxs = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1024 4096").split()]
    for ctx in ctxs:
        for B in batches:
            nb = B * (ctx // BS)
            kvs = [(torch.empty(nb, Hkv, D // 8, BS, 8, dtype=torch.float16, device=dev).uniform_(-0.1, 0.1),
                    torch.empty(nb, Hkv, D, BS, dtype=torch.float16, device=dev).uniform_(-0.1, 0.1)) for _ in range(NL)]
            bt = torch.randperm(nb, device=dev).to(torch.int32).reshape(B, -1)
            sl = torch.full((B, ), ctx, dtype=torch.int32, device=dev)
            q = torch.randn(B, H, D, dtype=torch.float16, device=dev) * 0.1
            out = torch.empty_like(q)
            P = (ctx + 511) // 512
            tmp = torch.empty(B, H, P, D, dtype=torch.float16, device=dev)
            es = torch.empty(B, H, P, dtype=torch.float32, device=dev)
            ml = torch.empty_like(es)

            def v1():
                for kc, vc in kvs:
                    ops.paged_attention_v1(out, q, kc, vc, Hkv, D**-0.5, bt, sl, BS, ctx, None, "auto", 1.0)

            def v2():
                for kc, vc in kvs:
                    ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, Hkv, D**-0.5, bt, sl, BS, ctx, None, "auto", 1.0)

            row = [f"ctx {ctx:5d} batch {B:3d}  v1 {time_graph(v1):7.2f} us"]
            for part in ("512", "256", "128", None):
                _lib.set_tuning("NMX_ATTN_PART", part)
                ps = _lib.lib().nmx_paged_attention_partition_size(B, H, Hkv, ctx)
                row.append(f"v2/{part or 'auto'}({ps}) {time_graph(v2):7.2f}")
            _lib.set_tuning("NMX_ATTN_PART", None)
            print("  ".join(row), flush=True)
            del kvs


if __name__ == "__main__":
    main()
