"""GPU parity tests for paged_attention_v1 / v2 (HIP, through the C-ABI) against the CPU oracle, the fp32 torch
restatement of the reference test's expected value, and the reference-made golden vectors.
Mirrors tests/kernels/test_attention.py:119-284 of the reference (tolerance atol 1e-3 / rtol 1e-5; fp8 KV atol 1e-2)."""
import random

import pytest
import torch

import oracle
from util import (DTYPES, create_kv_caches_with_random, from_bits, load_golden, ref_single_query_cached_kv_attention,
                  seed_all)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARTITION = 512


def run_hip(ops, version, q, kc, vc, kvh, scale, bt, sl, block_size, max_len, alibi, kv_dtype, kv_scale, **bs):
    qg = q.to(DEV)
    out = torch.full_like(qg, float("nan"))
    al = alibi.to(DEV) if alibi is not None else None
    if version == "v1":
        ops.paged_attention_v1(out, qg, kc.to(DEV), vc.to(DEV), kvh, scale, bt.to(DEV), sl.to(DEV), block_size, max_len,
                               al, kv_dtype, kv_scale, **bs)
    else:
        S, H, D = q.shape
        P = (max_len + PARTITION - 1) // PARTITION
        tmp = torch.empty(S, H, P, D, dtype=q.dtype, device=DEV)
        es = torch.empty(S, H, P, dtype=torch.float32, device=DEV)
        ml = torch.empty(S, H, P, dtype=torch.float32, device=DEV)
        ops.paged_attention_v2(out, es, ml, tmp, qg, kc.to(DEV), vc.to(DEV), kvh, scale, bt.to(DEV), sl.to(DEV),
                               block_size, max_len, al, kv_dtype, kv_scale, **bs)
    torch.cuda.synchronize()
    return out.cpu()


def run_oracle(version, q, kc, vc, kvh, scale, bt, sl, block_size, max_len, alibi, kv_dtype, kv_scale, **bs):
    out = torch.empty_like(q)
    if version == "v1":
        oracle.paged_attention_v1(out, q, kc, vc, kvh, scale, bt, sl, block_size, max_len, alibi, kv_dtype, kv_scale, **bs)
    else:
        S, H, D = q.shape
        P = (max_len + PARTITION - 1) // PARTITION
        tmp = torch.empty(S, H, P, D, dtype=q.dtype)
        es = torch.empty(S, H, P, dtype=torch.float32)
        ml = torch.empty(S, H, P, dtype=torch.float32)
        oracle.paged_attention_v2(out, es, ml, tmp, q, kc, vc, kvh, scale, bt, sl, block_size, max_len, alibi, kv_dtype,
                                  kv_scale, **bs)
    return out


@pytest.mark.parametrize("name", ["attn_bf16_gqa", "attn_bf16_alibi_mha", "attn_f32_gqa", "attn_bf16_opt125m"])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_attention_golden(ops, name, version):
    """Outputs of the reference's own CPU backend (bf16; the reference CPU backend has no fp16)."""
    g = load_golden(name)
    dt = DTYPES[str(g["dtype"])]
    q, kc, vc = from_bits(g["q"], dt), from_bits(g["k_cache"], dt), from_bits(g["v_cache"], dt)
    bt, sl = torch.from_numpy(g["block_tables"]), torch.from_numpy(g["seq_lens"])
    al = torch.from_numpy(g["alibi_slopes"]) if "alibi_slopes" in g.files else None
    out = run_hip(ops, version, q, kc, vc, int(g["num_kv_heads"]), float(g["scale"]), bt, sl, 16,
                  int(g["max_seq_len"]), al, "auto", 1.0)
    # bf16 outputs: one bf16 ulp at |x| <= 0.125 is 4.9e-4
    torch.testing.assert_close(out.float(), from_bits(g["out_" + version], dt).float(), atol=2e-3, rtol=1e-5)


NUM_BLOCKS = 1031
MAX_SEQ_LEN = 2100


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8), (8, 1)])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 192, 256])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("use_alibi", [False, True])
def test_paged_attention(ops, version, num_heads, head_size, block_size, dtype, use_alibi):
    if use_alibi and (head_size not in (64, 128) or block_size != 16):
        pytest.skip("alibi covered on a subset")
    seed_all(0)
    num_seqs = 7
    nq, nkv = num_heads
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [random.randint(1, MAX_SEQ_LEN) for _ in range(num_seqs)]
    seq_lens[-1] = MAX_SEQ_LEN
    seq_lens[0] = 1
    seq_lens[1] = 513
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, NUM_BLOCKS - 1) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(NUM_BLOCKS, block_size, 1, nkv, head_size, "auto", dtype)
    kc, vc = kcs[0], vcs[0]
    alibi = torch.randn(nq, dtype=torch.float32) if use_alibi else None
    out = run_hip(ops, version, q, kc, vc, nkv, scale, bt, sl, block_size, max_len, alibi, "auto", 1.0)
    ref = ref_single_query_cached_kv_attention(q, nq // nkv, kc, vc, bt, sl, scale, alibi)
    torch.testing.assert_close(out.float(), ref, atol=1e-3, rtol=1e-5)
    if head_size == 128 and block_size == 16:
        orc = run_oracle(version, q, kc, vc, nkv, scale, bt, sl, block_size, max_len, alibi, "auto", 1.0)
        torch.testing.assert_close(out.float(), orc.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("kv_dtype", ["fp8", "fp8_e5m2"])
def test_paged_attention_fp8_kv(ops, version, block_size, dtype, kv_dtype):
    seed_all(1)
    num_seqs, nq, nkv, head_size = 5, 32, 8, 128
    scale = float(head_size**-0.5)
    kv_scale = 0.75
    q = torch.empty(num_seqs, nq, head_size, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [700, 1, 33, 512, 1025]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, head_size, kv_dtype, dtype)
    out = run_hip(ops, version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    orc = run_oracle(version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    torch.testing.assert_close(out.float(), orc.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("head_size", [64, 80, 128, 256])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
def test_paged_attention_float32(ops, version, head_size, block_size, kv_dtype):
    """float32 queries (tests/kernels/test_attention.py DTYPES includes torch.float): fp32 cache or fp8 cache,
    alibi, GQA, ragged lengths across the 512-token partition boundary."""
    seed_all(5)
    num_seqs, nq, nkv = 5, 8, 2
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=torch.float32).uniform_(-scale, scale)
    seq_lens = [1, 511, 513, 1200, 77]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, head_size, kv_dtype, torch.float32)
    alibi = torch.randn(nq, dtype=torch.float32) * 0.1
    kv_scale = 0.5 if kv_dtype != "auto" else 1.0
    out = run_hip(ops, version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, alibi, kv_dtype, kv_scale)
    orc = run_oracle(version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, alibi, kv_dtype, kv_scale)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out, orc, atol=1e-3 if kv_dtype == "auto" else 1e-2, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("sliding_step", [0, 1, -1])
def test_paged_attention_blocksparse(ops, version, sliding_step):
    """Block-sparse path (attention_kernels.cu:209-256; Phi-3-small style: local 16 blocks, vertical stride 8)."""
    seed_all(2)
    num_seqs, nq, nkv, head_size, block_size = 4, 16, 4, 128, 16
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=torch.half).uniform_(-scale, scale)
    seq_lens = [1500, 64, 777, 2048]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 511) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(512, block_size, 1, nkv, head_size, "auto", torch.half)
    bs = dict(tp_rank=1, blocksparse_local_blocks=4, blocksparse_vert_stride=8, blocksparse_block_size=64,
              blocksparse_head_sliding_step=sliding_step)
    out = run_hip(ops, version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, "auto", 1.0, **bs)
    orc = run_oracle(version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, "auto", 1.0, **bs)
    torch.testing.assert_close(out.float(), orc.float(), atol=1e-3, rtol=1e-5)


def test_paged_attention_strided_query_and_nan_tail(ops):
    """query sliced out of a fused qkv tensor (q_stride != H*D) and NaNs parked past seq_len in the last block."""
    seed_all(3)
    num_seqs, nq, nkv, head_size, block_size = 3, 32, 8, 128, 16
    scale = float(head_size**-0.5)
    qkv = torch.empty(num_seqs, (nq + 2 * nkv) * head_size, dtype=torch.half).uniform_(-scale, scale)
    q = qkv[:, :nq * head_size].view(num_seqs, nq, head_size)
    seq_lens = [37, 300, 5]
    bt = torch.arange(num_seqs * 19, dtype=torch.int32).reshape(num_seqs, 19)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(64, block_size, 1, nkv, head_size, "auto", torch.half)
    kc, vc = kcs[0], vcs[0]
    for i, L in enumerate(seq_lens):  # poison everything past the end of each sequence's last block
        b = int(bt[i, (L - 1) // block_size])
        o = L % block_size
        if o:
            kc[b, :, :, o:, :] = float("nan")
            vc[b, :, :, o:] = float("nan")
    qg = qkv.to(DEV)[:, :nq * head_size].view(num_seqs, nq, head_size)
    out = torch.empty(num_seqs, nq, head_size, dtype=torch.half, device=DEV)
    ops.paged_attention_v1(out, qg, kc.to(DEV), vc.to(DEV), nkv, scale, bt.to(DEV), sl.to(DEV), block_size, 300, None, "auto", 1.0)
    ref = ref_single_query_cached_kv_attention(q, nq // nkv, kc, vc, bt, sl, scale, None)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.cpu().float(), ref, atol=1e-3, rtol=1e-5)


def test_paged_attention_rejects_bad_arguments(ops):
    q = torch.zeros(2, 8, 72, dtype=torch.half, device=DEV)
    kc = torch.zeros(4, 8, 9, 16, 8, dtype=torch.half, device=DEV)
    vc = torch.zeros(4, 8, 72, 16, dtype=torch.half, device=DEV)
    bt = torch.zeros(2, 4, dtype=torch.int32, device=DEV)
    sl = torch.ones(2, dtype=torch.int32, device=DEV)
    out = torch.zeros_like(q)
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        ops.paged_attention_v1(out, q, kc, vc, 8, 1.0, bt, sl, 16, 16, None, "auto", 1.0)
    q = torch.zeros(2, 8, 64, dtype=torch.half, device=DEV)
    out = torch.zeros_like(q)
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        ops.paged_attention_v1(out, q, kc, vc, 8, 1.0, bt, sl, 4, 16, None, "auto", 1.0)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.paged_attention_v1(out, q, kc, vc, 8, 1.0, bt, sl, 16, 16, None, "int3", 1.0)


def test_large_batch_property(ops):
    """BASELINE-size case (64 seqs x 1024 ctx, Llama-3-8B geometry): size-independent properties instead of the slow
    oracle — (i) v1 == v2, (ii) permuting the physical blocks (with the block table) leaves the output unchanged."""
    seed_all(4)
    S, nq, nkv, D, BS, L = 64, 32, 8, 128, 16, 1024
    scale = float(D**-0.5)
    NB = S * L // BS
    q = torch.empty(S, nq, D, dtype=torch.half, device=DEV).uniform_(-scale, scale)
    kc = torch.empty(NB, nkv, D // 8, BS, 8, dtype=torch.half, device=DEV).uniform_(-scale, scale)
    vc = torch.empty(NB, nkv, D, BS, dtype=torch.half, device=DEV).uniform_(-scale, scale)
    bt = torch.randperm(NB, device=DEV).to(torch.int32).reshape(S, L // BS)
    sl = torch.full((S, ), L, dtype=torch.int32, device=DEV)
    out1 = torch.empty_like(q)
    ops.paged_attention_v1(out1, q, kc, vc, nkv, scale, bt, sl, BS, L, None, "auto", 1.0)
    out2 = torch.empty_like(q)
    P = L // PARTITION
    tmp = torch.empty(S, nq, P, D, dtype=torch.half, device=DEV)
    es = torch.empty(S, nq, P, dtype=torch.float32, device=DEV)
    ml = torch.empty(S, nq, P, dtype=torch.float32, device=DEV)
    ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, nkv, scale, bt, sl, BS, L, None, "auto", 1.0)
    torch.testing.assert_close(out1.float(), out2.float(), atol=1e-3, rtol=1e-5)
    perm = torch.randperm(NB, device=DEV)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(NB, device=DEV)
    out3 = torch.empty_like(q)
    ops.paged_attention_v1(out3, q, kc[perm].contiguous(), vc[perm].contiguous(), nkv, scale,
                           inv[bt.long()].to(torch.int32), sl, BS, L, None, "auto", 1.0)
    assert torch.equal(out1, out3)
    # spot-check 2 sequences against the oracle
    idx = [0, 37]
    o = run_oracle("v1", q[idx].cpu(), kc.cpu(), vc.cpu(), nkv, scale, bt[idx].cpu(), sl[idx].cpu(), BS, L, None, "auto", 1.0)
    torch.testing.assert_close(out1[idx].cpu().float(), o.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("nq,nkv,head_size", [(32, 8, 128), (12, 12, 64), (40, 2, 128)])  # GQA 4, MHA, 20 heads per kv head (two q tiles)
def test_paged_attention_absmax(ops, version, kv_dtype, dtype, nq, nkv, head_size):
    """Round 3: paged_attention_v1/v2_absmax write the SAME output bits as the plain ops and partial maxima whose maximum is
    out.abs().max() exactly; scaled_fp8_quant_partials on them gives the codes and the scale of scaled_fp8_quant(out)."""
    seed_all(nq + head_size)
    block_size, num_seqs = 16, 7
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [700, 1, 33, 512, 1025, 513, 16]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, head_size, kv_dtype, dtype)
    kv_scale = 0.75 if kv_dtype != "auto" else 1.0
    plain = run_hip(ops, version, q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    qg, kc, vc, btg, slg = q.to(DEV), kcs[0].to(DEV), vcs[0].to(DEV), bt.to(DEV), sl.to(DEV)
    out = torch.full_like(qg, float("nan"))
    if version == "v1":
        amax = ops.paged_attention_v1_absmax(out, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None, kv_dtype, kv_scale)
        assert amax.numel() == num_seqs * nkv * ((nq // nkv + 15) // 16)
    else:
        P = (max_len + PARTITION - 1) // PARTITION
        tmp = torch.empty(num_seqs, nq, P, head_size, dtype=dtype, device=DEV)
        es = torch.empty(num_seqs, nq, P, dtype=torch.float32, device=DEV)
        ml = torch.empty(num_seqs, nq, P, dtype=torch.float32, device=DEV)
        amax = ops.paged_attention_v2_absmax(out, es, ml, tmp, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None,
                                             kv_dtype, kv_scale)
        assert amax.numel() == num_seqs * nq
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(torch.int16), plain.view(torch.int16))
    assert float(amax.max()) == float(out.float().abs().max())
    if version == "v2":  # one entry per (sequence, head)
        assert torch.equal(amax.view(num_seqs, nq), out.float().abs().amax(dim=-1))
    flat = out.view(num_seqs, nq * head_size)
    q1, s1 = ops.scaled_fp8_quant_partials(flat, amax)
    q0, s0 = ops.scaled_fp8_quant(flat)
    assert torch.equal(s0, s1) and torch.equal(q0.view(torch.uint8), q1.view(torch.uint8))


@pytest.mark.parametrize("part", ["64", "128", "256", None])
@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("num_seqs,nq,nkv,head_size", [(1, 32, 8, 128), (3, 12, 12, 64), (2, 40, 2, 128)])
def test_paged_attention_v2_fine_partitions(ops, tune, part, kv_dtype, dtype, num_seqs, nq, nkv, head_size):
    """Round 3 (late): at small batch paged_attention_v2 / v2_absmax split a sequence into partitions finer than the contract's
    512 tokens (own temporaries, nmx_paged_attention_v2_ps). Forced sizes (NMX_ATTN_PART) and the library's own choice
    (None: 1 - 3 sequences take 128- or 256-token partitions here) against the oracle's v2 and the fp32 restatement through the
    reference's bars; the absmax variant writes the same bits and exact maxima; the caller's temporaries stay untouched."""
    if dtype == torch.float and (kv_dtype != "auto" or part not in ("128", None)):
        pytest.skip("float32 queries covered on a subset")
    seed_all(nq + head_size + num_seqs)
    block_size = 16
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [1024, 129, 577][:num_seqs]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, head_size, kv_dtype, dtype)
    kv_scale = 0.75 if kv_dtype != "auto" else 1.0
    tune(NMX_ATTN_PART=part)
    from neuralmagic_vllm_amd import _lib
    ps = _lib.lib().nmx_paged_attention_partition_size(num_seqs, nq, nkv, max_len)
    assert ps == (int(part) if part is not None else ps) and ps < 512  # every case here takes the fine path
    qg, kc, vc, btg, slg = q.to(DEV), kcs[0].to(DEV), vcs[0].to(DEV), bt.to(DEV), sl.to(DEV)
    P = (max_len + PARTITION - 1) // PARTITION
    tmp = torch.full((num_seqs, nq, P, head_size), 7.0, dtype=dtype, device=DEV)
    es = torch.full((num_seqs, nq, P), 7.0, dtype=torch.float32, device=DEV)
    ml = torch.full((num_seqs, nq, P), 7.0, dtype=torch.float32, device=DEV)
    out = torch.full_like(qg, float("nan"))
    ops.paged_attention_v2(out, es, ml, tmp, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None, kv_dtype, kv_scale)
    torch.cuda.synchronize()
    assert bool((tmp == 7.0).all()) and bool((es == 7.0).all()) and bool((ml == 7.0).all())
    orc = run_oracle("v2", q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    atol = 1e-3 if kv_dtype == "auto" else 1e-2
    torch.testing.assert_close(out.cpu().float(), orc.float(), atol=atol, rtol=1e-5)
    if kv_dtype == "auto":
        ref = ref_single_query_cached_kv_attention(q, nq // nkv, kcs[0], vcs[0], bt, sl, scale, None)
        torch.testing.assert_close(out.cpu().float(), ref, atol=1e-3, rtol=1e-5)
    if dtype != torch.float:
        out2 = torch.full_like(qg, float("nan"))
        amax = ops.paged_attention_v2_absmax(out2, es, ml, tmp, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None,
                                             kv_dtype, kv_scale)
        torch.cuda.synchronize()
        assert torch.equal(out2.view(torch.int16), out.view(torch.int16))
        assert torch.equal(amax.view(num_seqs, nq), out.float().abs().amax(dim=-1))
    # the forced size through the C-ABI's checks
    if part == "64":
        tune(NMX_ATTN_PART="96")  # not a multiple of 64: the library answers the contract's 512
        assert _lib.lib().nmx_paged_attention_partition_size(num_seqs, nq, nkv, max_len) == 512


@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16])
@pytest.mark.parametrize("num_seqs,nq,nkv,head_size,part", [(1, 32, 8, 128, None), (4, 32, 8, 128, "128"), (3, 40, 2, 128, "64"),
                                                             (2, 12, 12, 64, "256")])
def test_paged_attention_v2_inkernel_reduce(ops, tune, monkeypatch, kv_dtype, dtype, num_seqs, nq, nkv, head_size, part):
    """Round 3 (late): with fine partitions the reduce runs inside the attention kernel - the last partition workgroup to
    arrive at a (sequence, kv head, q tile) reduces its heads (v2_last_arriver_reduce), no reduce launch. Same bits as the
    two-launch form (NMX_ATTN_INKERNEL_REDUCE=0: the reduce kernel, the same device function), counters back at zero, and the
    result does not change over 200 back-to-back launches with ragged sequence lengths (a stale read of another workgroup's
    partial would show as a flicker)."""
    from neuralmagic_vllm_amd import _custom_ops
    seed_all(nq + head_size + num_seqs)
    block_size = 16
    scale = float(head_size**-0.5)
    q = torch.empty(num_seqs, nq, head_size, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [1024, 129, 577, 1000][:num_seqs]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, head_size, kv_dtype, dtype)
    kv_scale = 0.75 if kv_dtype != "auto" else 1.0
    tune(NMX_ATTN_PART=part)
    qg, kc, vc, btg, slg = q.to(DEV), kcs[0].to(DEV), vcs[0].to(DEV), bt.to(DEV), sl.to(DEV)
    P = (max_len + PARTITION - 1) // PARTITION
    tmp = torch.empty(num_seqs, nq, P, head_size, dtype=dtype, device=DEV)
    es = torch.empty(num_seqs, nq, P, dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)

    def run(absmax=False):  # (qg: the binding at call time)
        out = torch.full_like(qg, float("nan"))
        if absmax:
            am = ops.paged_attention_v2_absmax(out, es, ml, tmp, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None, kv_dtype, kv_scale)
            return out, am
        ops.paged_attention_v2(out, es, ml, tmp, qg, kc, vc, nkv, scale, btg, slg, block_size, max_len, None, kv_dtype, kv_scale)
        return out, None

    monkeypatch.setenv("NMX_ATTN_INKERNEL_REDUCE", "0")
    two, two_am = run(absmax=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("NMX_ATTN_INKERNEL_REDUCE", "1")
    one, one_am = run(absmax=True)
    torch.cuda.synchronize()
    assert not torch.isnan(one.float()).any()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16)) and torch.equal(one_am, two_am)
    # back-to-back launches on ALTERNATING inputs (the temporaries come back at the same addresses: a stale read of another
    # workgroup's partial would return the other input's values), under load from a second stream that keeps the CUs' caches busy
    qg2 = torch.empty_like(qg).uniform_(-scale, scale)
    q_first = qg
    monkeypatch.setenv("NMX_ATTN_INKERNEL_REDUCE", "0")
    qg = qg2
    two_b = run()[0]
    torch.cuda.synchronize()
    monkeypatch.setenv("NMX_ATTN_INKERNEL_REDUCE", "1")
    assert not torch.equal(two_b.view(torch.int16), two.view(torch.int16))
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    outs = []
    for i in range(200):
        if i % 8 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        qg = q_first if i % 2 == 0 else qg2
        outs.append(run()[0])
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        want = two if i % 2 == 0 else two_b
        assert torch.equal(o.view(torch.int16), want.view(torch.int16)), f"launch {i}"
    qg = q_first
    for buf in _custom_ops._V2_COUNTERS.values():
        assert int(buf.abs().max()) == 0
    orc = run_oracle("v2", q, kcs[0], vcs[0], nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    torch.testing.assert_close(one.cpu().float(), orc.float(), atol=1e-3 if kv_dtype == "auto" else 1e-2, rtol=1e-5)
