"""Seeded random shapes through the DEFAULT dispatch of gptq_marlin_gemm (decode kernel, row-block kernel with and without
K splits, wide-tile kernel, large-tile kernel; padding workgroups, ragged row blocks, K that does not divide evenly over the
slices) and through the fused / deferred forms of the same GEMM. The reference test fixes a shape grid
(tests/kernels/test_marlin_gemm.py:126-179, covered by tests/test_marlin_gpu.py); dispatch here depends on (M, N, K) in
many more ways than the reference's tile table, so this file samples the space instead. The second half does the same for
cutlass_scaled_mm (fp8 / int8).

Expected value: a @ w_ref in fp32 with w_ref = fp16((q - 8) * s), the reference test's own expectation
(marlin_utils.py:72-109 builds w_ref the same way), bar compute_max_diff < 1e-3 (fp16) / 4e-3 (bf16), tighter than the 0.04 there. Weights are made on the GPU:
random 4-bit codes packed in the GPTQ checkpoint layout (quant_utils.py:125-146, element k at bits 4 (k % 8) of row k / 8) and
repacked by gptq_marlin_repack, which tests/test_marlin_gpu.py pins bit-exactly against the reference's packer."""
import random

import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(rng):
    M = rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 17, 24, 31, 32, 33, 47, 64, 65, 96, 100, 128, 129, 200, 256, 257, 300])
    N = 64 * rng.choice([1, 2, 3, 5, 8, 12, 16, 31, 33, 48, 64, 96, 100, 112, 128, 130, 224, 448])
    K = 128 * rng.choice([1, 2, 3, 4, 7, 8, 11, 16, 28, 32, 33, 56, 64, 72])
    group = rng.choice([-1, 128])
    return M, N, K, group


def _bar(dtype):
    """The north star's bar for the dequant GEMMs, mean|d| / mean|ref| <= 1e-3 (fp16; bf16 outputs carry 2^-9 of output
    rounding: 4e-3) - the same bar as every grid test, NOT the reference's 0.04 (tests/kernels/test_marlin_gemm.py:114)."""
    return 1e-3 if dtype == torch.float16 else 4e-3


def _make(M, N, K, group, dtype, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    q = torch.randint(0, 16, (K, N), dtype=torch.int32, device=DEV, generator=g)
    groups = 1 if group == -1 else K // group
    s = (torch.rand(groups, N, device=DEV, generator=g) * 0.01 + 0.002).to(dtype)
    w_ref = ((q - 8).to(dtype).view(groups, K // groups, N) * s[:, None, :]).view(K, N)  # (q - 8) exact, one rounding
    shifts = (4 * torch.arange(8, device=DEV, dtype=torch.int32)).view(1, 8, 1)
    packed = (q.view(K // 8, 8, N) << shifts).sum(dim=1, dtype=torch.int32)  # nibbles do not overlap: sum == or
    a = torch.randn(M, K, dtype=dtype, device=DEV, generator=g)
    return a, packed, s, w_ref


CASES = []
_rng = random.Random(20240)
while len(CASES) < 96:
    c = _case(_rng)
    if c not in CASES:
        CASES.append(c)


@pytest.mark.parametrize("M,N,K,group", CASES)
def test_random_shape_default_dispatch(ops, M, N, K, group):
    dtype = torch.float16 if (M + N // 64 + K // 128) % 3 else torch.bfloat16
    seed_all(M * 7 + N + K)
    a, packed, s, w_ref = _make(M, N, K, group, dtype, M + N + K)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(max(N // 64 * 16, 16), dtype=torch.int32, device=DEV)
    mq = ops.gptq_marlin_repack(packed, e, K, N, 4)
    ms = packing.marlin_permute_scales(s, K, N, group)
    out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    ref = (a.float() @ w_ref.float())
    torch.cuda.synchronize()
    assert compute_max_diff(out.float().cpu(), ref.cpu()) < _bar(dtype)
    # the deferred form: slabs summed in the reduce kernel's order -> the same bits
    d = ops.gptq_marlin_gemm_deferred(a, mq, ms, e, e, ws, 4, M, N, K, True)
    assert torch.equal(d.materialize().view(torch.int16), out.view(torch.int16))
    # gate | up halves + silu_and_mul as one op: the same bits as the two ops
    if N % 128 == 0:
        two = torch.empty(M, N // 2, dtype=dtype, device=DEV)
        ops.silu_and_mul(two, out)
        one = ops.gptq_marlin_gemm_silu_and_mul(a, mq, ms, e, e, ws, 4, M, N, K, True)
        torch.cuda.synchronize()
        assert torch.equal(one.view(torch.int16), two.view(torch.int16))


@pytest.mark.parametrize("M,N,K,group", CASES[:16])
def test_random_shape_oracle_slice(ops, M, N, K, group):
    """the same launches against the CPU oracle GEMM on a column slice (the oracle walks every weight: seconds at full size)"""
    a, packed, s, _ = _make(M, N, K, group, torch.float16, M + N + K)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(max(N // 64 * 16, 16), dtype=torch.int32, device=DEV)
    mq = ops.gptq_marlin_repack(packed, e, K, N, 4)
    ms = packing.marlin_permute_scales(s, K, N, group)
    out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).cpu()
    cols = min(N, 128)
    # the first `cols` columns are whole 64-column Marlin groups: 128 words per k-tile row and group, scales permuted
    # inside 64- (grouped) / 32-column (channel-wise) chunks
    mq_c = mq[:, :cols * 2].contiguous().cpu()
    ms_c = ms.cpu().reshape(-1, N // 64, 64)[:, :cols // 64].reshape(-1, cols).contiguous()
    orc = oracle.gptq_marlin_gemm(a.cpu(), mq_c, ms_c, None, None, None, 4, M, cols, K, True)
    assert compute_max_diff(out[:, :cols].float(), orc.float()) < _bar(torch.float16)


# ---- cutlass_scaled_mm (fp8 x fp8, int8 x int8): per-wave K-slice kernels (M <= 64), tile kernel with / without K splits ----
def _mm_case(rng):
    m = rng.choice([1, 3, 16, 17, 33, 64, 65, 100, 128, 130, 256, 300])
    n = 16 * rng.choice([1, 4, 5, 16, 31, 64, 69, 96, 256, 384, 1026])
    k = 128 * rng.choice([1, 2, 3, 8, 11, 32, 33, 64, 112])
    return m, n, k, rng.choice([True, False]), rng.choice([True, False]), rng.choice([True, False])


MM_CASES = []
while len(MM_CASES) < 48:
    c = _mm_case(_rng)
    if c not in MM_CASES:
        MM_CASES.append(c)


@pytest.mark.parametrize("m,n,k,is_fp8,per_token,per_channel", MM_CASES)
def test_random_shape_scaled_mm(ops, tune, m, n, k, is_fp8, per_token, per_channel):
    """default dispatch against the fp32 product on the device (test_cutlass.py:35-47's baseline and bars); int8
    accumulates exactly, so the tile kernel and the per-wave kernels must agree bit for bit where both apply; the deferred
    form + materialize equals the plain op bit for bit (fp8, per-tensor scales)."""
    g = torch.Generator(device=DEV)
    g.manual_seed(m + n + k)
    out_dtype = torch.bfloat16 if is_fp8 else torch.float16
    if is_fp8:
        a = torch.randn(m, k, device=DEV, generator=g).clamp(-448, 448).round().to(torch.float8_e4m3fn)
        b = torch.randn(n, k, device=DEV, generator=g).clamp(-448, 448).round().to(torch.float8_e4m3fn)
    else:
        a = (torch.randn(m, k, device=DEV, generator=g) * 5).clamp(-128, 127).round().to(torch.int8)
        b = (torch.randn(n, k, device=DEV, generator=g) * 5).clamp(-128, 127).round().to(torch.int8)
    sa = torch.rand((m, 1) if per_token else (1, 1), device=DEV, generator=g) / 10 + 0.01
    sb = torch.rand((1, n) if per_channel else (1, 1), device=DEV, generator=g) / 10 + 0.01
    out = ops.cutlass_scaled_mm(a, b.t(), sa, sb, out_dtype)
    base = (sa * (sb * torch.mm(a.float(), b.float().t()))).to(out_dtype)
    torch.testing.assert_close(out, base, rtol=1e-2, atol=5e-2 if is_fp8 else 1e-1)
    if not is_fp8:
        tune(NMX_MM_TILE="0")
        other = ops.cutlass_scaled_mm(a, b.t(), sa, sb, out_dtype)
        tune(NMX_MM_TILE=None)
        assert torch.equal(out, other)
    elif not per_token and not per_channel:
        plain = ops.cutlass_scaled_mm(a, b.t(), sa, sb, torch.float16)
        d = ops.cutlass_scaled_mm_deferred(a, b.t(), sa, sb, torch.float16)  # slabs live in the scratch: consume at once
        if d.splits > 1:  # the consumer ops apply sa * (sb * sum) and round once, like the reduce launch's epilogue
            res0 = torch.zeros(m, n, dtype=torch.float16, device=DEV)
            w1 = torch.ones(n, dtype=torch.float16, device=DEV)
            r_plain = res0.clone()
            x_plain = plain.clone()
            ops.fused_add_rms_norm(x_plain, r_plain, w1, 1e-5)
            r_fused = res0.clone()
            x_fused = ops.fused_add_rms_norm_splitk(d, r_fused, w1, 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(r_plain.view(torch.int16), r_fused.view(torch.int16))  # residual = the GEMM output itself
            assert torch.equal(x_plain.view(torch.int16), x_fused.view(torch.int16))
        else:
            assert torch.equal(d.out.view(torch.int16), plain.view(torch.int16))


# ---- paged_attention v1 / v2: head layouts, head sizes, block sizes and sequence-length mixes the fixed grids do not combine ----
def _attn_case(rng):
    nkv = rng.choice([1, 2, 4, 8])
    qpk = rng.choice([1, 2, 4, 5, 7, 8, 16, 17, 32])
    if nkv * qpk > 64:
        nkv = max(1, 64 // qpk)
    head = rng.choice([64, 80, 96, 112, 128, 192, 256])
    bs = rng.choice([8, 16, 32])
    dtype = rng.choice([torch.half, torch.bfloat16])
    kv = rng.choice(["auto", "auto", "fp8", "fp8_e5m2"])
    nseq = rng.choice([1, 2, 3, 7, 12])
    lens = [rng.choice([1, 2, 15, 16, 17, 31, 33, 100, 511, 512, 513, 777, 1024, 1025, 1536, 1537]) for _ in range(nseq)]
    return rng.choice(["v1", "v2"]), nkv * qpk, nkv, head, bs, dtype, kv, tuple(lens), rng.choice([False, False, True])


ATTN_CASES = []
while len(ATTN_CASES) < 64:
    c = _attn_case(_rng)
    if c not in ATTN_CASES:
        ATTN_CASES.append(c)


@pytest.mark.parametrize("version,nq,nkv,head,bs,dtype,kv,lens,alibi", ATTN_CASES)
def test_random_paged_attention(ops, version, nq, nkv, head, bs, dtype, kv, lens, alibi):
    """against the CPU oracle (attention_kernels.cu:86-669 restated), the bar of tests/kernels/test_attention.py:119-284"""
    from test_attention_gpu import run_hip, run_oracle
    from util import create_kv_caches_with_random
    seed_all(nq * 3 + head + len(lens))
    rnd = random.Random(nq + head + sum(lens))
    scale = float(head**-0.5)
    nseq, nblocks = len(lens), 300
    q = torch.empty(nseq, nq, head, dtype=dtype).uniform_(-scale, scale)
    max_len = max(lens)
    mb = (max_len + bs - 1) // bs
    bt = torch.tensor([[rnd.randint(0, nblocks - 1) for _ in range(mb)] for _ in range(nseq)], dtype=torch.int32)
    sl = torch.tensor(lens, dtype=torch.int32)
    kcs, vcs = create_kv_caches_with_random(nblocks, bs, 1, nkv, head, kv, dtype)
    slopes = torch.randn(nq, dtype=torch.float32) if alibi else None
    kv_scale = 0.6 if kv != "auto" else 1.0
    out = run_hip(ops, version, q, kcs[0], vcs[0], nkv, scale, bt, sl, bs, max_len, slopes, kv, kv_scale)
    orc = run_oracle(version, q, kcs[0], vcs[0], nkv, scale, bt, sl, bs, max_len, slopes, kv, kv_scale)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float(), orc.float(), atol=1e-2 if kv != "auto" else 2e-3, rtol=1e-5)


# ---- AWQ on the Marlin kernel and the checkpoint-layout awq_gemm: the two device paths of one checkpoint must agree ----
def _awq_case(rng):
    # awq_gemm checks OC % group_size == 0 like the reference (gemm_kernels.cu:505-513): N in multiples of 128; the Marlin
    # path needs >= 2 groups of 128
    return (rng.choice([1, 4, 8, 16, 17, 40, 64, 65, 128, 200, 256]), 128 * rng.choice([1, 2, 3, 8, 10, 32, 56, 64]),
            128 * rng.choice([2, 3, 5, 8, 28, 32, 64]))


AWQ_CASES = []
while len(AWQ_CASES) < 32:
    c = _awq_case(_rng)
    if c not in AWQ_CASES:
        AWQ_CASES.append(c)


@pytest.mark.parametrize("M,N,K", AWQ_CASES)
def test_random_shape_awq_paths(ops, M, N, K):
    """awq_marlin_gemm (load-time repack) == awq_gemm (checkpoint layout) to rounding, both == a @ ((q - z) * s) to the GEMM
    bar; the deferred form of the Marlin path equals its plain form bit for bit."""
    g = torch.Generator(device=DEV)
    g.manual_seed(M + N + K)
    groups = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=DEV, generator=g)
    qz = torch.randint(-2**31, 2**31 - 1, (groups, N // 8), dtype=torch.int32, device=DEV, generator=g)
    sc = (torch.rand(groups, N, device=DEV, generator=g) * 0.004 + 0.002).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV, generator=g)
    mq, ms, mz = ops.awq_marlin_repack(qw, qz, sc)
    marlin = ops.awq_marlin_gemm(a, mq, ms, mz, M, N, K)
    direct = ops.awq_gemm(a, qw, sc, qz, 8)
    w = ops.awq_dequantize(qw, sc, qz, 0, 0, 0)  # [K, N] fp16: (q - z) * s in the reference's arithmetic
    ref = a.float() @ w.float()
    torch.cuda.synchronize()
    assert compute_max_diff(marlin.float().cpu(), ref.cpu()) < _bar(torch.float16)
    assert compute_max_diff(direct.float().cpu(), ref.cpu()) < _bar(torch.float16)
    d = ops.awq_marlin_gemm_deferred(a, mq, ms, mz, M, N, K)
    assert torch.equal(d.materialize().view(torch.int16), marlin.view(torch.int16))


# ---- context_attention_fwd (prefill over paged context + new tokens): shared-LDS and per-wave kernels, every masking rule ----
def _prefill_case(rng):
    hkv = rng.choice([1, 2, 3, 4, 8])
    qpk = rng.choice([1, 2, 3, 4, 8])
    head = rng.choice([64, 80, 96, 112, 128, 192, 256])
    bs = rng.choice([8, 16, 32])
    dtype = rng.choice([torch.float16, torch.bfloat16])
    batch = rng.choice([1, 2, 5, 9])
    max_q = rng.choice([1, 7, 16, 63, 64, 65, 130, 300])
    max_ctx = rng.choice([0, 5, 16, 100, 257, 600])
    window = rng.choice([0, 0, 0, 8, 100])
    alibi = rng.choice([False, False, True])
    return hkv * qpk, hkv, head, bs, dtype, batch, max_q, max_ctx, window, alibi, rng.randint(0, 10**6)


PREFILL_CASES = []
while len(PREFILL_CASES) < 48:
    c = _prefill_case(_rng)
    if c not in PREFILL_CASES:
        PREFILL_CASES.append(c)


@pytest.mark.parametrize("H,Hkv,D,bs,dtype,batch,max_q,max_ctx,window,alibi,seed", PREFILL_CASES)
def test_random_prefill_attention(ops, H, Hkv, D, bs, dtype, batch, max_q, max_ctx, window, alibi, seed):
    """against the CPU oracle (prefix_prefill.py:674-812 restated; parity unpinned) with the bars of
    tests/test_prefill_gpu.py"""
    import test_prefill_gpu as P
    seed_all(seed)
    c = P.make_case(batch, H, Hkv, D, bs, dtype, max_q=max_q, max_ctx=max_ctx, cache_blocks=max(256, batch * ((max_ctx + bs - 1) // bs + 2)))
    slopes = (torch.rand(H) * 0.3) if alibi else None
    if alibi and window:
        window = 0  # the reference's alibi path has no sliding window (prefix_prefill.py:765-797)
    out = P.run_hip(c, alibi=slopes, window=window)
    orc = P.run_oracle(c, alibi=slopes, window=window)
    assert not torch.isnan(out).any()
    tol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float16 else dict(atol=1.5e-2, rtol=1.5e-2)
    torch.testing.assert_close(out.float(), orc.float(), **tol)


# ---- 8-bit GPTQ-Marlin and fp8-Marlin (W8A16) through the default dispatch ----
W8_CASES = []
while len(W8_CASES) < 40:
    c = _case(_rng)[:3] + (_rng.choice([-1, 128]), _rng.choice(["int8", "fp8"]))
    if c not in W8_CASES:
        W8_CASES.append(c)


@pytest.mark.parametrize("M,N,K,group,kind", W8_CASES)
def test_random_shape_w8a16(ops, M, N, K, group, kind):
    """gptq_marlin_gemm with 8-bit weights / fp8_marlin_gemm (channel-wise scales: fp8.py:258-287) against a @ w_ref, the
    reference test's expectation and bar (tests/kernels/test_marlin_gemm.py:126-179, 238-304)"""
    g = torch.Generator(device=DEV)
    g.manual_seed(M + N + K)
    dtype = torch.float16 if (M + K // 128) % 2 else torch.bfloat16
    a = torch.randn(M, K, dtype=dtype, device=DEV, generator=g)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(max(N // 64 * 16, 16), dtype=torch.int32, device=DEV)
    shifts = (8 * torch.arange(4, device=DEV, dtype=torch.int32)).view(1, 4, 1)
    if kind == "int8":
        q = torch.randint(0, 256, (K, N), dtype=torch.int32, device=DEV, generator=g)
        groups = 1 if group == -1 else K // group
        s = (torch.rand(groups, N, device=DEV, generator=g) * 0.001 + 0.0002).to(dtype)
        w_ref = ((q - 128).to(dtype).view(groups, K // groups, N) * s[:, None, :]).view(K, N)
        packed = (q.view(K // 4, 4, N) << shifts).sum(dim=1, dtype=torch.int32)
        mq = ops.gptq_marlin_repack(packed, e, K, N, 8)
        ms = packing.marlin_permute_scales(s, K, N, group)
        out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 8, M, N, K, True)
        d = ops.gptq_marlin_gemm_deferred(a, mq, ms, e, e, ws, 8, M, N, K, True)
        assert torch.equal(d.materialize().view(torch.int16), out.view(torch.int16))
    else:
        w8 = (torch.randn(K, N, device=DEV, generator=g) * 2).to(torch.float8_e4m3fn)
        q = w8.view(torch.uint8).to(torch.int32)
        packed = (q.view(K // 4, 4, N) << shifts).sum(dim=1, dtype=torch.int32)  # marlin_utils.py:226-247
        mq = ops.gptq_marlin_repack(packed, e, K, N, 8)
        s = (torch.rand(1, N, device=DEV, generator=g) * 0.01 + 0.002).to(dtype)
        ms = packing.marlin_permute_scales(s, K, N, -1)
        w_ref = (w8.to(dtype) * s)
        out = ops.fp8_marlin_gemm(a, mq, ms, ws, 8, M, N, K)
    ref = a.float() @ w_ref.float()
    torch.cuda.synchronize()
    assert compute_max_diff(out.float().cpu(), ref.cpu()) < _bar(dtype)


# ---- fused consumers of a deferred GEMM: rotary + KV-cache write, residual add + RMS norm ----
def _rope_case(rng):
    kvh = rng.choice([1, 2, 4, 8])
    qpk = rng.choice([1, 2, 4, 7])
    D = rng.choice([64, 96, 128, 256])
    while ((kvh * qpk + 2 * kvh) * D) % 64:
        kvh *= 2
    return (rng.choice([1, 3, 8, 16, 33, 64, 100, 256]), kvh * qpk, kvh, D, rng.choice([8, 16, 32]), rng.choice(["auto", "fp8"]),
            128 * rng.choice([2, 8, 32]))


ROPE_CASES = []
while len(ROPE_CASES) < 32:
    c = _rope_case(_rng)
    if c not in ROPE_CASES:
        ROPE_CASES.append(c)


@pytest.mark.parametrize("M,H,KVH,D,BS,kv_dtype,K", ROPE_CASES)
def test_random_rope_reshape_and_cache(ops, M, H, KVH, D, BS, kv_dtype, K):
    """qkv GEMM (deferred or not, as the dispatch decides) -> rotary + cache write in one launch == gptq_marlin_gemm ->
    rotary_embedding -> reshape_and_cache, bit for bit (pos_encoding_kernels.cu:10-96, cache_kernels.cu:153-278)"""
    g = torch.Generator(device=DEV)
    g.manual_seed(M + H + D + K)
    N, NB = (H + 2 * KVH) * D, 64
    mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV, generator=g)
    ms = (torch.rand(K // 128, N, device=DEV, generator=g) * 0.004 + 0.002).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV, generator=g)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    plain = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    cos_sin = torch.randn(512, D, device=DEV, generator=g).half()
    positions = torch.randint(0, 512, (M, ), device=DEV, generator=g)
    slots = torch.randperm(NB * BS, device=DEV, generator=g)[:M].long() if M <= NB * BS else None
    if M > 2:
        slots[1] = -1
    cdt = torch.uint8 if kv_dtype == "fp8" else torch.float16
    x = 16 if kv_dtype == "fp8" else 8
    kc0 = torch.randint(0, 100, (NB, KVH, D // x, BS, x), device=DEV, generator=g).to(cdt)
    vc0 = torch.randint(0, 100, (NB, KVH, D, BS), device=DEV, generator=g).to(cdt)
    kv_scale = 0.05 if kv_dtype == "fp8" else 1.0
    kc_a, vc_a = kc0.clone(), vc0.clone()
    qa, ka, va = plain.split([H * D, KVH * D, KVH * D], dim=-1)
    ops.rotary_embedding(positions, qa, ka, D, cos_sin, True)
    ops.reshape_and_cache(ka.view(-1, KVH, D), va.view(-1, KVH, D), kc_a, vc_a, slots, kv_dtype, kv_scale)
    kc_b, vc_b = kc0.clone(), vc0.clone()
    d = ops.gptq_marlin_gemm_deferred(a, mq, ms, e, e, ws, 4, M, N, K, True)
    qkv = ops.rope_reshape_and_cache(positions, d, H, KVH, D, cos_sin, kc_b, vc_b, slots, kv_dtype, kv_scale)
    torch.cuda.synchronize()
    assert torch.equal(qkv.view(torch.int16), plain.view(torch.int16))
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)


@pytest.mark.parametrize("M,N,K", [(m, 64 * n, 128 * k) for m, n, k in
                                   [(1, 16, 32), (3, 33, 8), (16, 64, 32), (17, 5, 64), (64, 64, 112), (65, 128, 8), (100, 31, 33),
                                    (256, 64, 32), (300, 16, 112), (8, 224, 16), (33, 96, 28), (128, 112, 56)]])
def test_random_add_rms_norm_consumer(ops, M, N, K):
    g = torch.Generator(device=DEV)
    g.manual_seed(M + N + K)
    mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV, generator=g)
    ms = (torch.rand(K // 128, N, device=DEV, generator=g) * 0.004 + 0.002).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV, generator=g)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    plain = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    res0 = torch.randn(M, N, dtype=torch.float16, device=DEV, generator=g)
    w = (torch.rand(N, device=DEV, generator=g) + 0.5).half()
    res_a, res_b = res0.clone(), res0.clone()
    ops.fused_add_rms_norm(plain, res_a, w, 1e-5)
    d = ops.gptq_marlin_gemm_deferred(a, mq, ms, e, e, ws, 4, M, N, K, True)
    fused = ops.fused_add_rms_norm_splitk(d, res_b, w, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.int16), plain.view(torch.int16)) and torch.equal(res_a.view(torch.int16), res_b.view(torch.int16))


# ---- gptq_marlin_24_gemm (2:4-sparse): row-block kernel (M <= 32 and the shapes the wide rules leave) and marlin_wide_kernel<SP> ----
def _sparse_case(rng):
    M = rng.choice([1, 5, 16, 17, 32, 33, 48, 64, 65, 96, 128, 129, 200, 256, 257, 300])
    N = 128 * rng.choice([1, 2, 3, 8, 12, 16, 33, 48, 56, 64, 112, 224])
    K = 128 * rng.choice([1, 2, 3, 4, 7, 8, 16, 28, 32, 33, 56])
    return M, N, K, rng.choice([-1, 128]), rng.choice([4, 4, 8])


SPARSE_CASES = []
_srng = random.Random(20243)
while len(SPARSE_CASES) < 40:
    c = _sparse_case(_srng)
    if c not in SPARSE_CASES:
        SPARSE_CASES.append(c)


@pytest.mark.parametrize("M,N,K,group,bits", SPARSE_CASES)
def test_random_shape_sparse24(ops, tune, M, N, K, group, bits):
    """Default dispatch of gptq_marlin_24_gemm on random compressed words + random VALID metadata against the CPU oracle on the
    first and last 128 columns (1e-3), against the row-block kernel on the whole output, and its deferred form bit for bit."""
    gen = torch.Generator().manual_seed(M + N + K + bits)
    pack = 32 // bits
    mq = torch.randint(-2**31, 2**31 - 1, (K // 32, N * 16 // pack), dtype=torch.int32, generator=gen)
    nib = torch.tensor([0x4, 0x8, 0xC, 0x9, 0xD, 0xE], dtype=torch.int32)  # (idx0 | idx1 << 2), idx0 < idx1
    pick = nib[torch.randint(0, 6, (K // 32, N * 2, 4), generator=gen)]
    meta = (pick[..., 0] | (pick[..., 1] << 4) | (pick[..., 2] << 8) | (pick[..., 3] << 12)).to(torch.int16)
    groups = 1 if group == -1 else K // group
    ms = (torch.rand(groups, N, generator=gen) * 0.01 + 0.005).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, generator=gen)
    ws = torch.zeros(N // 128 * 64, dtype=torch.int32, device=DEV)
    ad, qd, md, sd = a.to(DEV), mq.to(DEV), meta.to(DEV), ms.to(DEV)
    c = ops.gptq_marlin_24_gemm(ad, qd, md, sd, ws, bits, M, N, K)
    for lo in sorted({0, N - 128}):
        orc = oracle.gptq_marlin_24_gemm(a, mq[:, lo * 16 // pack:(lo + 128) * 16 // pack].contiguous(), meta[:, lo * 2:(lo + 128) * 2].contiguous(),
                                         ms[:, lo:lo + 128].contiguous(), None, bits, M, 128, K)
        assert compute_max_diff(c[:, lo:lo + 128].float().cpu(), orc) < 1e-3, (M, N, K, group, bits, lo)
    d = ops.gptq_marlin_24_gemm_deferred(ad, qd, md, sd, ws, bits, M, N, K)
    assert torch.equal(d.materialize().view(torch.int16), c.view(torch.int16))
    tune(NMX_GEMM_WIDE="0")
    base = ops.gptq_marlin_24_gemm(ad, qd, md, sd, ws, bits, M, N, K)
    assert compute_max_diff(c.float().cpu(), base.float().cpu()) < 1e-3
