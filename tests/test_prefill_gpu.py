"""GPU parity tests for context_attention_fwd (prefill over paged context + new tokens) against the CPU oracle and the
fp32 torch restatement of the reference test's expected value (tests/kernels/test_prefix_prefill.py)."""
import random

import pytest
import torch

import oracle
from util import ref_prefix_prefill, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_case(batch, H, Hkv, D, block_size, dtype, max_q, max_ctx, min_len=1, cache_blocks=256):
    query_lens = [random.randint(min_len, max_q) for _ in range(batch)]
    ctx_lens = [random.randint(0, max_ctx) for _ in range(batch)]
    ctx_lens[0] = 0  # a pure prefill
    seq_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    T = sum(query_lens)
    q = torch.empty(T, H, D).uniform_(-1, 1).to(dtype)
    k = torch.empty(T, Hkv, D).uniform_(-1, 1).to(dtype)
    v = torch.empty(T, Hkv, D).uniform_(-1, 1).to(dtype)
    k_cache = torch.empty(cache_blocks, Hkv, D // 8, block_size, 8).uniform_(-1, 1).to(dtype)
    v_cache = torch.empty(cache_blocks, Hkv, D, block_size).uniform_(-1, 1).to(dtype)
    # poison the slots past each context so that masking is exercised
    max_blocks = (max(max_ctx, 1) + block_size - 1) // block_size + 1
    perm = torch.randperm(cache_blocks)[:batch * max_blocks].reshape(batch, max_blocks).to(torch.int32)
    for b in range(batch):
        c = ctx_lens[b]
        if c % block_size:
            blk = int(perm[b, c // block_size])
            k_cache[blk, :, :, c % block_size:, :] = float("nan")
            v_cache[blk, :, :, c % block_size:] = float("nan")
    b_start = torch.cumsum(torch.tensor([0] + query_lens[:-1]), 0).to(torch.int32)
    return dict(q=q, k=k, v=v, k_cache=k_cache, v_cache=v_cache, b_loc=perm, b_start_loc=b_start,
                b_seq_len=torch.tensor(seq_lens, dtype=torch.int32), b_ctx_len=torch.tensor(ctx_lens, dtype=torch.int32),
                max_input_len=max(query_lens))


def run_hip(c, alibi=None, window=None):
    from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd
    g = {n: (t.to(DEV) if torch.is_tensor(t) else t) for n, t in c.items()}
    o = torch.zeros_like(g["q"])
    context_attention_fwd(g["q"], g["k"], g["v"], o, g["k_cache"], g["v_cache"], g["b_loc"], g["b_start_loc"],
                          g["b_seq_len"], g["b_ctx_len"], g["max_input_len"], None if alibi is None else alibi.to(DEV), window)
    return o.cpu()


def run_oracle(c, alibi=None, window=None):
    o = torch.zeros_like(c["q"])
    oracle.context_attention_fwd(c["q"], c["k"], c["v"], o, c["k_cache"], c["v_cache"], c["b_loc"], c["b_start_loc"],
                                 c["b_seq_len"], c["b_ctx_len"], c["max_input_len"], alibi, window)
    return o


@pytest.mark.parametrize("heads", [(8, 8), (8, 2), (8, 1)])
@pytest.mark.parametrize("head_size", [128, 96, 64])
@pytest.mark.parametrize("sliding_window", [0, 16, 100])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_contexted_kv_attention(ops, heads, head_size, sliding_window, dtype):
    """tests/kernels/test_prefix_prefill.py:28-220 (MHA / GQA / MQA, sliding windows)."""
    seed_all(0)
    c = make_case(6, heads[0], heads[1], head_size, 16, dtype, max_q=150, max_ctx=200)
    out = run_hip(c, window=sliding_window)
    orc = run_oracle(c, window=sliding_window)
    ref = ref_prefix_prefill(c["q"], c["k"], c["v"], c["k_cache"], c["v_cache"], c["b_loc"], c["b_start_loc"],
                             c["b_seq_len"], c["b_ctx_len"], None, sliding_window)
    assert not torch.isnan(out).any()
    tol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float16 else dict(atol=1.5e-2, rtol=1.5e-2)
    torch.testing.assert_close(out.float(), orc.float(), **tol)
    torch.testing.assert_close(out.float(), ref, **tol)


@pytest.mark.parametrize("gq", ["1", "2", "11", "12", "14"])
@pytest.mark.parametrize("heads", [(8, 2), (8, 1), (6, 3)])
@pytest.mark.parametrize("head_size", [128, 80, 64])
def test_two_heads_per_wave_shape(ops, tune, gq, heads, head_size):
    """Every launch shape of the kernel - per-wave K / V loads with one / two query heads of a kv head per wave ("1", "2")
    and the workgroup-shared LDS tiles with one / two / four heads per wave ("11", "12", "14"; head sizes that are a
    multiple of 64, other sizes fall back) - on the same ragged case, with alibi, plus a long prompt that takes the
    default route."""
    seed_all(3)
    tune(NMX_PREFILL_GQ=gq)
    c = make_case(5, heads[0], heads[1], head_size, 16, torch.float16, max_q=200, max_ctx=150)
    alibi = torch.rand(heads[0]) * 0.2
    out = run_hip(c, alibi=alibi)
    torch.testing.assert_close(out.float(), run_oracle(c, alibi=alibi).float(), atol=2e-3, rtol=2e-3)
    tune(NMX_PREFILL_GQ=None)
    c = make_case(2, heads[0], heads[1], head_size, 16, torch.float16, max_q=400, max_ctx=60, min_len=300)
    torch.testing.assert_close(run_hip(c).float(), run_oracle(c).float(), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_long_ragged_prompts(ops, dtype):
    """Many key tiles per workgroup, ragged lengths in one launch (a 1 500-token prompt over 700 cached tokens next to short
    ones): the double-buffered LDS tiles, the block-table prefetch across the cached / new boundary and the
    heaviest-first grid order of the shared kernel, GQA 8 : 2, head 128."""
    seed_all(7)
    random.seed(7)
    c = make_case(4, 8, 2, 128, 16, dtype, max_q=1500, max_ctx=700, min_len=1400, cache_blocks=512)
    # make the batch ragged: one long, one medium, two short sequences
    lens = [1500, 300, 65, 1]
    ctxs = [700, 0, 333, 15]
    T = sum(lens)
    for n in ("q", "k", "v"):
        c[n] = c[n][:T]
    c["b_start_loc"] = torch.tensor([0, 1500, 1800, 1865], dtype=torch.int32)
    c["b_seq_len"] = torch.tensor([a + b for a, b in zip(lens, ctxs)], dtype=torch.int32)
    c["b_ctx_len"] = torch.tensor(ctxs, dtype=torch.int32)
    c["max_input_len"] = 1500
    for n in ("k_cache", "v_cache"):  # make_case poisoned slots for ITS context lengths; ours differ
        c[n] = torch.nan_to_num(c[n].float(), nan=0.25).to(dtype)
    out = run_hip(c)
    orc = run_oracle(c)
    assert not torch.isnan(out).any()
    tol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float16 else dict(atol=1.5e-2, rtol=1.5e-2)
    torch.testing.assert_close(out.float(), orc.float(), **tol)


@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("head_size", [80, 112, 192, 256])
def test_contexted_kv_attention_alibi_and_shapes(ops, block_size, head_size):
    """alibi variant (_fwd_kernel_alibi, prefix_prefill.py:437-672), every block size, the remaining head sizes, strided
    q / k / v sliced out of one fused qkv tensor."""
    seed_all(1)
    H, Hkv = 4, 2
    c = make_case(4, H, Hkv, head_size, block_size, torch.float16, max_q=70, max_ctx=90)
    T = c["q"].shape[0]
    qkv = torch.zeros(T, (H + 2 * Hkv) * head_size, dtype=torch.float16)
    qkv[:, :H * head_size] = c["q"].reshape(T, -1)
    qkv[:, H * head_size:(H + Hkv) * head_size] = c["k"].reshape(T, -1)
    qkv[:, (H + Hkv) * head_size:] = c["v"].reshape(T, -1)
    c2 = dict(c)
    c2["q"] = qkv[:, :H * head_size].view(T, H, head_size)
    c2["k"] = qkv[:, H * head_size:(H + Hkv) * head_size].view(T, Hkv, head_size)
    c2["v"] = qkv[:, (H + Hkv) * head_size:].view(T, Hkv, head_size)
    alibi = torch.tensor([2**-(i + 1) for i in range(H)], dtype=torch.float32)
    gq = qkv.to(DEV)
    g = {n: (t.to(DEV) if torch.is_tensor(t) else t) for n, t in c.items()}
    g["q"] = gq[:, :H * head_size].view(T, H, head_size)
    g["k"] = gq[:, H * head_size:(H + Hkv) * head_size].view(T, Hkv, head_size)
    g["v"] = gq[:, (H + Hkv) * head_size:].view(T, Hkv, head_size)
    from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd
    o = torch.zeros(T, H, head_size, dtype=torch.float16, device=DEV)
    context_attention_fwd(g["q"], g["k"], g["v"], o, g["k_cache"], g["v_cache"], g["b_loc"].long(), g["b_start_loc"].long(),
                          g["b_seq_len"].long(), g["b_ctx_len"].long(), c["max_input_len"], alibi.to(DEV))
    ref = ref_prefix_prefill(c["q"], c["k"], c["v"], c["k_cache"], c["v_cache"], c["b_loc"], c["b_start_loc"],
                             c["b_seq_len"], c["b_ctx_len"], alibi, 0)
    torch.testing.assert_close(o.cpu().float(), ref, atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(o.cpu().float(), run_oracle(c2, alibi=alibi).float(), atol=2e-3, rtol=2e-3)


def test_forward_prefix_shim_and_errors(ops):
    from neuralmagic_vllm_amd.attention.ops.paged_attn import PagedAttention
    from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd
    seed_all(2)
    c = make_case(3, 8, 2, 128, 16, torch.float16, max_q=40, max_ctx=64)
    g = {n: (t.to(DEV) if torch.is_tensor(t) else t) for n, t in c.items()}
    qsl = torch.cat([g["b_start_loc"], torch.tensor([c["q"].shape[0]], dtype=torch.int32, device=DEV)])
    out = PagedAttention.forward_prefix(g["q"], g["k"], g["v"], g["k_cache"], g["v_cache"], g["b_loc"], qsl, g["b_seq_len"],
                                        g["b_ctx_len"], c["max_input_len"], None, None)
    torch.testing.assert_close(out.cpu().float(), run_oracle(c).float(), atol=2e-3, rtol=2e-3)
    with pytest.raises(RuntimeError, match="unsupported head size"):
        z = torch.zeros(4, 2, 24, dtype=torch.float16, device=DEV)
        context_attention_fwd(z, z, z, z.clone(), torch.zeros(4, 2, 3, 16, 8, dtype=torch.float16, device=DEV),
                              torch.zeros(4, 2, 24, 16, dtype=torch.float16, device=DEV), g["b_loc"][:1], g["b_start_loc"][:1],
                              g["b_seq_len"][:1], g["b_ctx_len"][:1], 4)


@pytest.mark.parametrize("heads", [(32, 8), (8, 8), (8, 1)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_plain_prefill_shim_ctx0(ops, heads, dtype):
    """ROCmFlashAttentionImpl's prefix-free prefill (rocm_flash_attn.py:359-394) through the shims of
    attention/ops/triton_flash_attention.py: `triton_attention` / `flash_attn_varlen_func` signatures on nmx_context_attention_fwd
    with an empty paged context; checked against the CPU oracle and the fp32 torch restatement (causal, GQA / MQA)."""
    from neuralmagic_vllm_amd.attention.ops.triton_flash_attention import flash_attn_varlen_func, triton_attention
    seed_all(11)
    H, Hkv = heads
    D = 128
    lens = [1, 37, 300, 64, 129]
    T = sum(lens)
    q = torch.empty(T, H, D).uniform_(-1, 1).to(dtype)
    k = torch.empty(T, Hkv, D).uniform_(-1, 1).to(dtype)
    v = torch.empty(T, Hkv, D).uniform_(-1, 1).to(dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    out, aux = triton_attention(q.to(DEV), k.to(DEV), v.to(DEV), None, cu.to(DEV), cu.to(DEV), max(lens), max(lens), True, D**-0.5, None)
    assert aux is None
    # expected value: the same problem as a contexted prefill with ctx = 0
    kc = torch.zeros(1, Hkv, D // 8, 16, 8, dtype=dtype)
    vc = torch.zeros(1, Hkv, D, 16, dtype=dtype)
    b_loc = torch.zeros(len(lens), 1, dtype=torch.int32)
    sl = torch.tensor(lens, dtype=torch.int32)
    ctx = torch.zeros(len(lens), dtype=torch.int32)
    ref = ref_prefix_prefill(q, k, v, kc, vc, b_loc, cu[:-1], sl, ctx)
    tol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float16 else dict(atol=1.5e-2, rtol=1.5e-2)
    torch.testing.assert_close(out.cpu().float(), ref, **tol)
    orc = torch.zeros_like(q)
    oracle.context_attention_fwd(q, k, v, orc, kc, vc, b_loc, cu[:-1].contiguous(), sl, ctx, max(lens), None, None)
    torch.testing.assert_close(out.cpu().float(), orc.float(), **tol)
    out2 = flash_attn_varlen_func(q.to(DEV), k.to(DEV), v.to(DEV), cu_seqlens_q=cu.to(DEV), cu_seqlens_k=cu.to(DEV), max_seqlen_q=max(lens),
                                  max_seqlen_k=max(lens), softmax_scale=D**-0.5, causal=True)
    assert torch.equal(out2, out)
    with pytest.raises(RuntimeError, match="causal"):
        triton_attention(q.to(DEV), k.to(DEV), v.to(DEV), None, cu.to(DEV), cu.to(DEV), max(lens), max(lens), False, D**-0.5, None)
    with pytest.raises(RuntimeError, match="alibi_slopes"):
        triton_attention(q.to(DEV), k.to(DEV), v.to(DEV), None, cu.to(DEV), cu.to(DEV), max(lens), max(lens), True, D**-0.5,
                         torch.zeros(1, device=DEV))
