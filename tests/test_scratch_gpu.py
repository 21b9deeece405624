"""Split-K scratch lifetime: per-call allocations, graph-private under capture,
owned by a DeferredGemm until it is consumed. Also: the `Tensor! out` ops of torch_bindings write their operands in place."""
import pytest
import torch

from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _quant(K, N, seed):
    seed_all(seed)
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, q, s, _, _, _ = packing.marlin_quantize(w, 4, 128, False)
    return w_ref.float(), q.to(DEV), s.to(DEV)


def test_scratch_is_per_call_and_graph_safe(ops):
    """Round 3: split-K scratch is a per-call allocation (no module-level buffer pinned per stream). A captured graph's
    scratch comes from the graph's private pool: replaying it after later eager calls of other sizes and after allocator
    churn still gives the right result and writes into nobody else's memory; eager calls leave nothing allocated behind;
    a DeferredGemm owns its slabs (a later GEMM on the same stream cannot clobber them)."""
    assert not hasattr(ops, "_scratch") and not hasattr(ops, "_retired")
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    K1, N1, M1 = 4096, 512, 16   # decode-sized GEMM with cross-workgroup K splits (partials go through the scratch)
    w1, q1, s1 = _quant(K1, N1, 1)
    a1 = torch.randn(M1, K1, dtype=torch.float16, device=DEV)
    ws1 = torch.zeros(N1 // 64 * 16, dtype=torch.int32, device=DEV)
    ref1 = a1.float().cpu() @ w1
    assert ops._lib.lib().nmx_marlin_gemm_scratch_bytes(M1, N1, K1) > 0, "pick a shape that splits K"
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        out = ops.gptq_marlin_gemm(a1, q1, s1, e, e, ws1, 4, M1, N1, K1, True)
        stream.synchronize()
        assert compute_max_diff(out.cpu(), ref1) < 1e-3
        del out
        base = torch.cuda.memory_allocated()
        for _ in range(3):
            ops.gptq_marlin_gemm(a1, q1, s1, e, e, ws1, 4, M1, N1, K1, True)
        stream.synchronize()
        assert torch.cuda.memory_allocated() == base  # nothing pinned behind the calls
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            out1 = ops.gptq_marlin_gemm(a1, q1, s1, e, e, ws1, 4, M1, N1, K1, True)
        g.replay()
        stream.synchronize()
        assert compute_max_diff(out1.cpu(), ref1) < 1e-3
        # a bigger problem on the same stream, then allocator churn: the graph's slabs must be untouched by either
        K2, N2, M2 = 4096, 2048, 64
        w2, q2, s2 = _quant(K2, N2, 2)
        a2 = torch.randn(M2, K2, dtype=torch.float16, device=DEV)
        ws2 = torch.zeros(N2 // 64 * 16, dtype=torch.int32, device=DEV)
        d1 = ops.gptq_marlin_gemm_deferred(a1, q1, s1, e, e, ws1, 4, M1, N1, K1, True)   # slabs pending ...
        out2 = ops.gptq_marlin_gemm(a2, q2, s2, e, e, ws2, 4, M2, N2, K2, True)           # ... while another GEMM runs
        stream.synchronize()
        assert compute_max_diff(out2.cpu(), a2.float().cpu() @ w2) < 1e-3
        assert d1.splits > 1 and compute_max_diff(d1.materialize().cpu(), ref1) < 1e-3
        junk = [torch.full((1 << 18,), 7.0, device=DEV) for _ in range(8)]
        out1.zero_()
        g.replay()
        stream.synchronize()
        assert compute_max_diff(out1.cpu(), ref1) < 1e-3
        assert all(bool((j == 7.0).all()) for j in junk)  # and the replay wrote into nobody else's memory
    ops.release_scratch()


def test_out_operands_written_in_place(ops):
    import neuralmagic_vllm_amd.torch_bindings  # noqa: F401  (registers torch.ops._C.*)
    seed_all(3)
    m, k, n = 32, 256, 128
    a = (torch.randn(m, k, device=DEV) * 0.5).to(torch.float8_e4m3fn)
    b = (torch.randn(n, k, device=DEV) * 0.5).to(torch.float8_e4m3fn).t()
    sa = torch.full((1,), 0.5, device=DEV)
    sb = torch.full((1,), 0.25, device=DEV)
    out = torch.full((m, n), float("nan"), dtype=torch.float16, device=DEV)
    ptr = out.data_ptr()
    torch.ops._C.cutlass_scaled_mm(out, a, b, sa, sb, None)
    ref = (a.float() @ b.float()) * 0.125
    assert out.data_ptr() == ptr and torch.allclose(out.float(), ref, atol=5e-2, rtol=2e-2)
    x = torch.randn(m, k, dtype=torch.float16, device=DEV)
    q = torch.zeros(m, k, dtype=torch.float8_e4m3fn, device=DEV)
    scale = torch.zeros(1, device=DEV)
    torch.ops._C.dynamic_scaled_fp8_quant(q, x, scale)
    q_ref, s_ref = ops.scaled_fp8_quant(x, None)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(scale, s_ref)
    qi = torch.zeros(m, k, dtype=torch.int8, device=DEV)
    si = torch.zeros(m, 1, device=DEV)
    torch.ops._C.dynamic_scaled_int8_quant(qi, x, si)
    qi_ref, si_ref = ops.scaled_int8_quant(x, None)
    assert torch.equal(qi, qi_ref) and torch.equal(si, si_ref)
    st = torch.full((1,), 0.05, device=DEV)
    torch.ops._C.static_scaled_int8_quant(qi, x, st)
    assert torch.equal(qi, ops.scaled_int8_quant(x, st)[0])
    torch.ops._C.static_scaled_fp8_quant(q, x, st)
    assert torch.equal(q.view(torch.uint8), ops.scaled_fp8_quant(x, st)[0].view(torch.uint8))
