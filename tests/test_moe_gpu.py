"""GPU tests of the MoE path: topk_softmax and moe_align_block_size against torch restatements of
csrc/moe/topk_softmax_kernels.cu / csrc/moe_align_block_size_kernels.cu, fused_moe against the reference test's own
torch_moe (tests/kernels/test_moe.py:20-37, atol 1e-2), and the fp8 method (fp8.py:382-560) against the same experts
evaluated from the dequantised weights."""
import pytest
import torch

from util import seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def torch_moe(a, w1, w2, score, topk, renormalize=False):
    b, d = a.shape
    a2 = a.view(b, -1, d).repeat(1, topk, 1).reshape(-1, d)
    out = torch.zeros(b * topk, w2.shape[1], dtype=torch.float32, device=a.device)
    score = torch.softmax(score.float(), dim=-1)
    topk_weight, topk_ids = torch.topk(score, topk)
    if renormalize:
        topk_weight = topk_weight / topk_weight.sum(dim=-1, keepdim=True)
    topk_weight, topk_ids = topk_weight.view(-1), topk_ids.view(-1)
    for i in range(w1.shape[0]):
        mask = topk_ids == i
        if mask.sum():
            h = a2[mask].float() @ w1[i].float().t()
            n = h.shape[1] // 2
            h = torch.nn.functional.silu(h[:, :n]) * h[:, n:]
            out[mask] = h.to(a.dtype).float() @ w2[i].float().t()
    return (out.view(b, -1, w2.shape[1]) * topk_weight.view(b, -1, 1)).sum(dim=1)


@pytest.mark.parametrize("m,e,topk", [(1, 8, 2), (33, 8, 2), (222, 64, 6), (7, 160, 6)])
def test_topk_softmax(ops, m, e, topk):
    seed_all(m)
    g = torch.randn(m, e, device=DEV)
    w = torch.empty(m, topk, device=DEV)
    ids = torch.empty(m, topk, dtype=torch.int32, device=DEV)
    src = torch.empty(m, topk, dtype=torch.int32, device=DEV)
    ops.topk_softmax(w, ids, src, g)
    rw, rid = torch.topk(torch.softmax(g, dim=-1), topk)
    assert torch.equal(ids.long(), rid)
    torch.testing.assert_close(w, rw, atol=1e-6, rtol=1e-5)
    assert torch.equal(src.cpu(), (torch.arange(topk)[None, :] * m + torch.arange(m)[:, None]).int())


@pytest.mark.parametrize("m,e,topk,block", [(1, 8, 2, 16), (50, 8, 2, 16), (300, 64, 6, 64), (40, 8, 2, 1)])
def test_moe_align_block_size(ops, m, e, topk, block):
    seed_all(m)
    ids = torch.randint(0, e, (m, topk), dtype=torch.int32, device=DEV)
    numel = m * topk
    max_sorted = numel + e * (block - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=DEV)
    expert_ids = torch.full(((max_sorted + block - 1) // block, ), -1, dtype=torch.int32, device=DEV)
    post = torch.empty(1, dtype=torch.int32, device=DEV)
    ops.moe_align_block_size(ids, e, block, sorted_ids, expert_ids, post)
    flat = ids.flatten().cpu()
    exp_sorted, exp_experts = [], []
    for ex in range(e):
        idx = (flat == ex).nonzero().flatten().tolist()
        pad = (-len(idx)) % block
        exp_sorted += idx + [numel] * pad
        exp_experts += [ex] * ((len(idx) + pad) // block)
    assert int(post) == len(exp_sorted)
    assert sorted_ids[:len(exp_sorted)].cpu().tolist() == exp_sorted
    assert expert_ids[:len(exp_experts)].cpu().tolist() == exp_experts


@pytest.mark.parametrize("m", [1, 33, 222])
@pytest.mark.parametrize("n,k", [(256, 128), (1024, 512)])
@pytest.mark.parametrize("e,topk", [(8, 2), (64, 6)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_moe(ops, m, n, k, e, topk, dtype):
    from neuralmagic_vllm_amd.layers.fused_moe import fused_moe
    seed_all(0)
    a = torch.randn(m, k, device=DEV, dtype=dtype) / 10
    w1 = torch.randn(e, 2 * n, k, device=DEV, dtype=dtype) / 10
    w2 = torch.randn(e, k, n, device=DEV, dtype=dtype) / 10
    score = torch.randn(m, e, device=DEV, dtype=dtype)
    out = fused_moe(a, w1, w2, score, topk, renormalize=False)
    ref = torch_moe(a, w1, w2, score, topk)
    assert torch.allclose(out.float(), ref, atol=1e-2, rtol=0)  # the reference test's bar (test_moe.py:58)


@pytest.mark.parametrize("scheme,serialized", [("dynamic", False), ("dynamic", True), ("static", True)])
def test_fp8_moe_method(ops, scheme, serialized):
    from neuralmagic_vllm_amd.layers.quantization.fp8 import Fp8Config, Fp8MoEMethod
    seed_all(1)
    e, n, k, m, topk = 8, 256, 512, 40, 2
    dtype = torch.float16
    layer = torch.nn.Module()
    method = Fp8MoEMethod(Fp8Config(is_checkpoint_fp8_serialized=serialized, activation_scheme=scheme))
    method.create_weights(layer, e, k, n, dtype)
    w13 = torch.randn(e, 2 * n, k, dtype=dtype) / 10
    w2 = torch.randn(e, k, n, dtype=dtype) / 10
    if serialized:
        # checkpoint: fp8 weights, separate scales for w1 / w3
        s13 = torch.stack([w13[:, :n].abs().amax(dim=(1, 2)), w13[:, n:].abs().amax(dim=(1, 2))], dim=1).float() / 448.0
        s2 = w2.abs().amax(dim=(1, 2)).float() / 448.0
        q13 = torch.cat([(w13[:, :n].float() / s13[:, 0, None, None]), (w13[:, n:].float() / s13[:, 1, None, None])], dim=1).to(torch.float8_e4m3fn)
        q2 = (w2.float() / s2[:, None, None]).to(torch.float8_e4m3fn)
        layer.w13_weight.data, layer.w2_weight.data = q13.to(DEV), q2.to(DEV)
        layer.w13_scale.data, layer.w2_scale.data = s13.to(DEV), s2.to(DEV)
        w13_eff = torch.cat([q13[:, :n].float() * s13[:, 0, None, None], q13[:, n:].float() * s13[:, 1, None, None]], dim=1)
        w2_eff = q2.float() * s2[:, None, None]
        if scheme == "static":
            layer.a13_scale.data = torch.full((e, ), 0.02, device=DEV)
            layer.a2_scale.data = torch.full((e, ), 0.05, device=DEV)
    else:
        layer.w13_weight.data, layer.w2_weight.data = w13.to(DEV), w2.to(DEV)
        layer.w13_scale.data, layer.w2_scale.data = layer.w13_scale.data.to(DEV), layer.w2_scale.data.to(DEV)
        w13_eff, w2_eff = w13.float(), w2.float()
    method.process_weights_after_loading(layer)
    assert layer.w13_weight.dtype == torch.float8_e4m3fn and layer.w13_scale.shape == (e, )
    x = (torch.randn(m, k, dtype=dtype) / 10).to(DEV)
    logits = torch.randn(m, e, dtype=dtype, device=DEV)
    out = method.apply(layer, x.clone(), logits, topk, renormalize=True)
    ref = torch_moe(x, w13_eff.to(DEV), w2_eff.to(DEV), logits, topk, renormalize=True)
    # four e4m3 operands in the chain (x, w13, the intermediate, w2), each with up to 2^-4 relative rounding (~3 % rms):
    # ~6 % mean relative error against the unquantised evaluation is the format, not the kernels (the fp8 GEMM itself is
    # held to baseline_scaled_mm in tests/test_quant_gpu.py)
    err = float((out.float() - ref).abs().mean() / ref.abs().mean())
    assert err < 1e-1, err


@pytest.mark.parametrize("m,e,topk,block", [(1, 8, 2, 16), (33, 8, 2, 16), (222, 8, 2, 64), (40, 64, 6, 32)])
@pytest.mark.parametrize("n,k", [(256, 128), (1000, 512)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_moe_scaled_mm(ops, m, e, topk, block, n, k, dtype):
    """The grouped fp8 GEMM against a per-pair torch evaluation: gather through sorted_token_ids, per-expert weight scale,
    optional routing weight, scatter to the pair id; padding slots and blocks past num_tokens_post_padded do nothing."""
    seed_all(m + n)
    numel = m * topk
    ids = torch.stack([torch.randperm(e)[:topk] for _ in range(m)]).int().to(DEV)
    a = (torch.randn(m, k, device=DEV) * 2).to(torch.float8_e4m3fn)
    w = (torch.randn(e, n, k, device=DEV) * 2).to(torch.float8_e4m3fn)
    a_s = torch.tensor([0.03], device=DEV)
    w_s = (torch.rand(e, device=DEV) * 0.05 + 0.01)
    tw = torch.rand(m, topk, device=DEV)
    max_sorted = numel + e * (block - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=DEV)
    expert_ids = torch.zeros((max_sorted + block - 1) // block, dtype=torch.int32, device=DEV)
    post = torch.empty(1, dtype=torch.int32, device=DEV)
    ops.moe_align_block_size(ids, e, block, sorted_ids, expert_ids, post)
    for use_tw, div in ((False, topk), (True, topk)):
        out = torch.full((numel, n), 7.0, dtype=dtype, device=DEV)
        ops.moe_scaled_mm(out, a, w, a_s, w_s, tw if use_tw else None, sorted_ids, expert_ids, post, div, block)
        flat = ids.flatten().long()
        rows = torch.arange(numel, device=DEV) // div
        ref = torch.einsum("rk,rnk->rn", a.float()[rows], w.float()[flat])
        if use_tw:
            ref = ref * tw.flatten()[:, None]
        ref = (ref * a_s * w_s[flat][:, None]).to(dtype)
        torch.testing.assert_close(out.float(), ref.float(), rtol=1e-2, atol=2e-2)


def test_fused_moe_fp8_is_graph_capturable(ops):
    """The fp8 MoE layer issues device launches only: it can be captured and replayed with new inputs."""
    from neuralmagic_vllm_amd.layers.fused_moe import fused_experts, fused_topk
    seed_all(3)
    e, n, k, m, topk = 8, 256, 512, 24, 2
    w1 = (torch.randn(e, 2 * n, k, device=DEV) / 4).to(torch.float8_e4m3fn)
    w2 = (torch.randn(e, k, n, device=DEV) / 4).to(torch.float8_e4m3fn)
    s1 = torch.full((e, ), 0.02, device=DEV)
    s2 = torch.full((e, ), 0.03, device=DEV)
    x = torch.randn(m, k, dtype=torch.float16, device=DEV) / 10
    logits = torch.randn(m, e, device=DEV)

    def run():
        tw, ids = fused_topk(x, logits, topk, True)
        return fused_experts(x, w1, w2, tw, ids, use_fp8=True, w1_scale=s1, w2_scale=s2)

    eager = run().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = run()
        out.zero_()
        g.replay()
        side.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("m,e,topk,block", [(1, 8, 2, 16), (33, 8, 2, 16), (222, 8, 2, 64), (40, 64, 6, 32)])
@pytest.mark.parametrize("n,k", [(256, 128), (1000, 448)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_moe_mm_unquantised(ops, m, e, topk, block, n, k, dtype):
    """Round 3: the unquantised grouped GEMM (the reference's Triton fused_moe_kernel with use_fp8 = False, fused_moe.py:20-292)
    against a per-pair fp32 torch evaluation: gather through sorted_token_ids, optional routing weight, scatter to the pair id;
    padding slots and blocks past num_tokens_post_padded do nothing (the canary value survives nowhere a row is valid)."""
    seed_all(m + n)
    numel = m * topk
    ids = torch.stack([torch.randperm(e)[:topk] for _ in range(m)]).int().to(DEV)
    a = (torch.randn(m, k, device=DEV) / 4).to(dtype)
    w = (torch.randn(e, n, k, device=DEV) / 4).to(dtype)
    tw = torch.rand(m, topk, device=DEV)
    max_sorted = numel + e * (block - 1)
    sorted_ids = torch.empty(max_sorted, dtype=torch.int32, device=DEV)
    expert_ids = torch.zeros((max_sorted + block - 1) // block, dtype=torch.int32, device=DEV)
    post = torch.empty(1, dtype=torch.int32, device=DEV)
    ops.moe_align_block_size(ids, e, block, sorted_ids, expert_ids, post)
    for use_tw in (False, True):
        out = torch.full((numel, n), 7.0, dtype=dtype, device=DEV)
        ops.moe_mm(out, a, w, tw if use_tw else None, sorted_ids, expert_ids, post, topk, block)
        flat = ids.flatten().long()
        rows = torch.arange(numel, device=DEV) // topk
        ref = torch.einsum("rk,rnk->rn", a.float()[rows], w.float()[flat])
        if use_tw:
            ref = ref * tw.flatten()[:, None]
        tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)
        torch.testing.assert_close(out.float(), ref.to(dtype).float(), **tol)


def test_fused_moe_unquantised_is_graph_capturable(ops):
    """The fp16 MoE layer is device launches only since round 3 (no host read of the expert histogram): capture + replay."""
    from neuralmagic_vllm_amd.layers.fused_moe import fused_experts, fused_topk
    seed_all(4)
    e, n, k, m, topk = 8, 256, 512, 24, 2
    w1 = (torch.randn(e, 2 * n, k, device=DEV) / 10).half()
    w2 = (torch.randn(e, k, n, device=DEV) / 10).half()
    x = torch.randn(m, k, dtype=torch.float16, device=DEV) / 10
    logits = torch.randn(m, e, device=DEV)

    def run():
        tw, ids = fused_topk(x, logits, topk, True)
        return fused_experts(x, w1, w2, tw, ids)

    eager = run().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = run()
        out.zero_()
        g.replay()
        side.synchronize()
    assert torch.equal(out, eager)
