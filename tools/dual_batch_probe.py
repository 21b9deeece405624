"""Experiment (round 3, late): does the decode step of a batch run faster as TWO micro-batches on two streams inside one HIP
graph - one micro-batch's HBM-bound paged attention under the other's matrix-bound GEMMs - than as one batch? The micro-batches
share the weights and are independent chains (no cross dependency): the graph has two parallel branches.
usage (GPU box): python3 tools/dual_batch_probe.py [batch] > gpurun_out/dual_batch.txt   (ONE batch per process: a second configuration in
the same process did not come back within 7 minutes on the box - not investigated)
result (batch 256): one batch 11.083 ms | two micro-batches of 128: one stream 14.052 ms, two streams 11.879 ms"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from neuralmagic_vllm_amd import _custom_ops as ops  # noqa: E402

dev = torch.device("cuda", 0)


def build(batch, share=None):
    var = bench.VARIANTS["int4"]
    cfg = dict(var["model"])
    m = bench.Llama3Decode(ops, cfg, batch, 1024, cfg["layers"], dev, variant="int4") if share is None else None
    if share is not None:
        # same weights, own tokens / KV cache / block tables: construct with one layer, then point at the shared weights
        cfg1 = dict(cfg)
        m = bench.Llama3Decode(ops, cfg1, batch, 1024, cfg["layers"], dev, variant="int4")
        m.layers, m.lm_head, m.embed, m.final_ln = share.layers, share.lm_head, share.embed, share.final_ln
    m.fuse = True
    return m


def timed(fn, steps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    for B in [int(a) for a in sys.argv[1:2]] or [256]:
        one = build(B)
        one.step()
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            one.step()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g1):
            one.step()
        t_one = timed(g1.replay)
        del g1
        kv_keep = one.kv
        one.kv = None
        torch.cuda.empty_cache()
        a = build(B // 2, share=one)
        b = build(B // 2, share=one)
        a.step(); b.step()
        torch.cuda.synchronize()
        # serial: both micro-batches on one stream
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs):
            a.step(); b.step()
        t_serial = timed(gs.replay)
        del gs
        # parallel branches: fork a side stream inside the capture
        side = torch.cuda.Stream()
        gp = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gp):
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                b.step()
            a.step()
            cur.wait_stream(side)
        t_par = timed(gp.replay)
        print(f"batch {B}: one batch {t_one:.3f} ms ({B / t_one * 1e3:.0f} tok/s) | two micro-batches of {B // 2}: serial {t_serial:.3f} ms, "
              f"two streams {t_par:.3f} ms ({B / t_par * 1e3:.0f} tok/s)", flush=True)
        del gp, a, b, one
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
