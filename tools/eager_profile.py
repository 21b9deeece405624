"""Host-side cost of the eager (no HIP graph) decode step: cProfile over a few steps of bench.py's Llama3Decode at a small batch.
usage (GPU box): python3 tools/eager_profile.py [batch]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    from neuralmagic_vllm_amd import _custom_ops as ops
    m = bench.Llama3Decode(ops, dict(bench.LLAMA3_8B), batch, 1024, 32, "cuda:0", variant="int4")
    m.fuse = True
    for _ in range(3):
        m.step_fused()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        m.step_fused()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
