"""Fused decode-path ops (extensions): the split-K reduction of gptq_marlin_gemm deferred into the consumer
(fused_add_rms_norm / silu_and_mul / rotary + reshape_and_cache). Each fused path must be BIT-IDENTICAL to the plain op
sequence on the HIP ops, and equal to the CPU oracle's unfused element-wise op applied to the same GEMM output
(reference ops: layernorm_kernels.cu:258-291, activation_kernels.cu:12-30, pos_encoding_kernels.cu:10-96,
cache_kernels.cu:153-278)."""
import pytest
import torch

import oracle
from util import seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(t):
    return t.contiguous().view({1: torch.uint8, 2: torch.int16, 4: torch.int32}[t.element_size()])


def _weights(K, N, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    q = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV, generator=g)
    s = (torch.rand(K // 128, N, device=DEV, generator=g) * 0.004 + 0.002).to(torch.float16)
    return q, s


def _gemm_pair(ops, a, q, s, K, N):
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    plain = ops.gptq_marlin_gemm(a, q, s, e, e, ws, 4, a.shape[0], N, K, True)
    g = ops.gptq_marlin_gemm_deferred(a, q, s, e, e, ws, 4, a.shape[0], N, K, True)
    return plain, g


@pytest.mark.parametrize("M", [1, 8, 16, 33, 64, 256])
@pytest.mark.parametrize("K,N", [(4096, 4096), (14336, 4096)])
def test_deferred_gemm_add_rms_norm(ops, M, K, N):
    seed_all(M)
    q, s = _weights(K, N, 1)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain, g = _gemm_pair(ops, a, q, s, K, N)
    gemm_out = plain.clone()  # the reduced GEMM output the plain sequence sees
    if M <= 64:
        assert g.splits > 1, "these shapes split K at small M: the fused form must be the one under test"
    res0 = torch.randn(M, N, dtype=torch.float16, device=DEV)
    w = (torch.rand(N, device=DEV) + 0.5).half()
    res_a, res_b = res0.clone(), res0.clone()
    ops.fused_add_rms_norm(plain, res_a, w, 1e-5)
    fused = ops.fused_add_rms_norm_splitk(g, res_b, w, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(_bits(fused), _bits(plain)) and torch.equal(_bits(res_a), _bits(res_b))
    # oracle's unfused op on the same GEMM output
    x_o, r_o = gemm_out.cpu(), res0.cpu()
    oracle.fused_add_rms_norm(x_o, r_o, w.cpu(), 1e-5)
    assert torch.equal(_bits(res_b.cpu()), _bits(r_o))  # the residual add is exact arithmetic
    # the sum of squares runs in a different order on the CPU: compare to rounding (same bar as tests/test_elementwise.py)
    torch.testing.assert_close(fused.cpu().float(), x_o.float(), atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("M", [1, 16, 48, 64])
def test_deferred_gemm_silu_and_mul(ops, M):
    K, N = 4096, 28672
    seed_all(M)
    q, s = _weights(K, N, 2)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain, g = _gemm_pair(ops, a, q, s, K, N)
    out_a = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
    out_b = torch.empty_like(out_a)
    ops.silu_and_mul(out_a, plain)
    ops.silu_and_mul_splitk(out_b, g)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out_a), _bits(out_b))
    out_o = torch.empty(M, N // 2, dtype=torch.float16)
    oracle.act_and_mul(out_o, plain.cpu(), "silu")
    # expf differs in the last bit between the GPU and libm: compare to rounding (as tests/test_elementwise.py does)
    torch.testing.assert_close(out_b.cpu().float(), out_o.float(), atol=2e-3, rtol=2e-3)


def _gate_up_case(ops, M, K, N, dtype, grouped, seed, bits=4):
    """(two-op result, one-op result) of silu_and_mul(gptq_marlin_gemm(a, w)) on the same operands"""
    seed_all(seed)
    q, s = _weights(K, N, seed)
    if bits == 8:  # twice the packed words per k-tile row; scales sized for |q - 128| <= 128
        q = torch.cat([q, q.flip(1)], dim=1).contiguous()
        s = (s / 16).to(s.dtype)
    if not grouped:
        s = s[:1].contiguous()
    s = s.to(dtype)
    a = torch.randn(M, K, dtype=dtype, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    plain = ops.gptq_marlin_gemm(a, q, s, e, e, ws, bits, M, N, K, True)
    two = torch.empty(M, N // 2, dtype=dtype, device=DEV)
    ops.silu_and_mul(two, plain)
    one = ops.gptq_marlin_gemm_silu_and_mul(a, q, s, e, e, ws, bits, M, N, K, True)
    torch.cuda.synchronize()
    return plain, two, one


@pytest.mark.parametrize("M", [1, 5, 8, 12, 16, 24, 32, 33, 64, 100, 128, 256, 300])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gate_up_gemm_silu_and_mul(ops, M, dtype):
    """gate_up projection + silu_and_mul as one op on the Llama-3-8B shape, default dispatch: M in (32, 256] takes the
    wide-tile kernel without a K split = the activation runs in the GEMM epilogue; the other sizes take GEMM + consumer.
    Bit-identical to the two ops, and the oracle's silu_and_mul of the same GEMM output to rounding."""
    K, N = 4096, 28672
    plain, two, one = _gate_up_case(ops, M, K, N, dtype, True, M)
    assert torch.equal(_bits(two), _bits(one))
    out_o = torch.empty(M, N // 2, dtype=dtype)
    oracle.act_and_mul(out_o, plain.cpu(), "silu")
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    torch.testing.assert_close(one.cpu().float(), out_o.float(), atol=tol, rtol=tol)


@pytest.mark.parametrize("wide", ["1,4,1", "2,2,1"])
def test_gate_up_gemm_silu_and_mul_int8(ops, tune, wide):
    """8-bit weights through the wide kernel's fused epilogue (the 64-row shapes are 4-bit only)"""
    K, N, M = 1024, 2 * 1280, 200
    tune(NMX_GEMM_WIDE=wide)
    _, two, one = _gate_up_case(ops, M, K, N, torch.float16, True, 12, bits=8)
    assert torch.isfinite(one.float()).all() and torch.equal(_bits(two), _bits(one))


@pytest.mark.parametrize("wide", ["1,4,1", "1,2,1", "2,2,1", "2,4,1", "1,2,1,4", "1,4,1,4", "1,4,2"])
@pytest.mark.parametrize("grouped", [True, False])
def test_gate_up_gemm_silu_and_mul_tiles(ops, tune, wide, grouped):
    """every tile shape of the wide kernel with the fused epilogue (forced through NMX_GEMM_WIDE; "1,4,2" splits K and must
    fall back to GEMM + consumer), ragged M, a column count that leaves padding workgroups"""
    K, N = 1024, 2 * 1280
    M = 50 if wide.endswith(",4") else 200
    tune(NMX_GEMM_WIDE=None)
    ref = _gate_up_case(ops, M, K, N, torch.float16, grouped, 11)[1]
    tune(NMX_GEMM_WIDE=wide)
    _, two, one = _gate_up_case(ops, M, K, N, torch.float16, grouped, 11)
    assert torch.equal(_bits(two), _bits(one))
    torch.testing.assert_close(one.float(), ref.float(), atol=2e-3, rtol=2e-3)  # other tile shape: another summation order


@pytest.mark.parametrize("M", [1, 7, 16, 64, 200])
@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("deferred", [True, False])
def test_rope_reshape_and_cache(ops, M, kv_dtype, deferred):
    H, KVH, D, BS, NB = 32, 8, 128, 16, 64
    K, N = 4096, (H + 2 * KVH) * D
    seed_all(M + 100)
    q, s = _weights(K, N, 3)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain, g = _gemm_pair(ops, a, q, s, K, N)
    raw = plain.cpu()  # the reduced, unrotated GEMM output
    if not deferred:
        g = plain.clone()
    max_pos = 512
    cos_sin = torch.randn(max_pos, D, device=DEV).half()
    positions = torch.randint(0, max_pos, (M, ), device=DEV)
    slots = torch.randperm(NB * BS, device=DEV)[:M].long()
    if M > 4:
        slots[3] = -1  # padding token: rotated like every row, not cached
    cdt = torch.uint8 if kv_dtype == "fp8" else torch.float16
    x = 16 if kv_dtype == "fp8" else 8
    kc0 = torch.randint(0, 100, (NB, KVH, D // x, BS, x), device=DEV).to(cdt)
    vc0 = torch.randint(0, 100, (NB, KVH, D, BS), device=DEV).to(cdt)
    kv_scale = 0.05 if kv_dtype == "fp8" else 1.0
    # plain sequence
    kc_a, vc_a = kc0.clone(), vc0.clone()
    qa, ka, va = plain.split([H * D, KVH * D, KVH * D], dim=-1)
    ops.rotary_embedding(positions, qa, ka, D, cos_sin, True)
    ops.reshape_and_cache(ka.view(-1, KVH, D), va.view(-1, KVH, D), kc_a, vc_a, slots, kv_dtype, kv_scale)
    # fused
    kc_b, vc_b = kc0.clone(), vc0.clone()
    qkv = ops.rope_reshape_and_cache(positions, g, H, KVH, D, cos_sin, kc_b, vc_b, slots, kv_dtype, kv_scale)
    torch.cuda.synchronize()
    assert torch.equal(_bits(qkv), _bits(plain))
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    # oracle's unfused ops on the same (unrotated) GEMM output
    qo, ko, vo = (t.contiguous() for t in raw.split([H * D, KVH * D, KVH * D], dim=-1))
    oracle.rotary_embedding(positions.cpu(), qo, ko, D, cos_sin.cpu(), True)
    kc_o, vc_o = kc0.cpu().clone(), vc0.cpu().clone()
    oracle.reshape_and_cache(ko.reshape(-1, KVH, D), vo.reshape(-1, KVH, D), kc_o, vc_o, slots.cpu(), kv_dtype, kv_scale)
    # hipcc contracts x*c - y*s into mixed-precision FMAs: one ulp of fp16 against the CPU's step-by-step rounding (the
    # same bar as tests/test_elementwise.py::test_rotary_embedding); the v heads and the V cache are pure moves
    torch.testing.assert_close(qkv.cpu()[:, :(H + KVH) * D].float(), torch.cat([qo, ko], dim=-1).float(), atol=2e-3, rtol=2e-3)
    assert torch.equal(vc_b.cpu(), vc_o)
    if kv_dtype == "auto":
        torch.testing.assert_close(kc_b.cpu().float(), kc_o.float(), atol=2e-3, rtol=2e-3)


def test_deferred_materialize_matches_plain(ops):
    K, N, M = 4096, 4096, 16
    seed_all(9)
    q, s = _weights(K, N, 4)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain, g = _gemm_pair(ops, a, q, s, K, N)
    assert g.splits > 1
    assert torch.equal(_bits(g.materialize()), _bits(plain))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,hidden", [(1, 4096), (64, 4096), (7, 1000), (256, 14336)])
def test_absmax_producers_and_quant_from_partials(ops, dtype, T, hidden):
    """rms_norm / fused_add_rms_norm / silu_and_mul with the per-token |max| side output (the producers of an fp8 linear
    layer's input): outputs identical to the plain ops, maxima exact, and scaled_fp8_quant_partials returns the codes and the
    scale of dynamic scaled_fp8_quant bit for bit."""
    seed_all(T + hidden)
    x = (torch.randn(T, hidden, dtype=dtype, device=DEV) * 3)
    w = torch.randn(hidden, dtype=dtype, device=DEV)
    out_a, out_b = torch.empty_like(x), torch.empty_like(x)
    amax = ops.rms_norm_absmax(out_a, x, w, 1e-5)
    ops.rms_norm(out_b, x, w, 1e-5)
    assert torch.equal(out_a, out_b)
    assert torch.equal(amax, out_b.float().abs().amax(dim=1))
    q_a, s_a = ops.scaled_fp8_quant_partials(out_a, amax)
    q_b, s_b = ops.scaled_fp8_quant(out_b)
    assert torch.equal(s_a, s_b) and torch.equal(q_a.view(torch.uint8), q_b.view(torch.uint8))
    # fused add
    res_a = torch.randn(T, hidden, dtype=dtype, device=DEV)
    res_b, xa, xb = res_a.clone(), x.clone(), x.clone()
    amax = ops.fused_add_rms_norm_absmax(xa, res_a, w, 1e-5)
    ops.fused_add_rms_norm(xb, res_b, w, 1e-5)
    assert torch.equal(xa, xb) and torch.equal(res_a, res_b)
    assert torch.equal(amax, xb.float().abs().amax(dim=1))
    # gated activation
    if hidden % 2 == 0:
        act_a = torch.empty(T, hidden // 2, dtype=dtype, device=DEV)
        act_b = torch.empty_like(act_a)
        amax = ops.silu_and_mul_absmax(act_a, x)
        ops.silu_and_mul(act_b, x)
        assert torch.equal(act_a, act_b)
        assert torch.equal(amax, act_b.float().abs().amax(dim=1))
        q_a, s_a = ops.scaled_fp8_quant_partials(act_a, amax)
        q_b, s_b = ops.scaled_fp8_quant(act_b)
        assert torch.equal(s_a, s_b) and torch.equal(q_a.view(torch.uint8), q_b.view(torch.uint8))


@pytest.mark.parametrize("M", [1, 16, 64, 256])
@pytest.mark.parametrize("K,N", [(8192, 1280), (1024, 8192), (3584, 8192)])
def test_deferred_awq_marlin_gemm(ops, M, K, N):
    """AWQ repacked onto the Marlin kernel (Llama-3-70B TP = 8 rank shapes): the deferred form + fused consumer is
    bit-identical to awq_marlin_gemm + the plain op."""
    seed_all(M + K)
    g128 = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=DEV)
    qz = torch.randint(-2**31, 2**31 - 1, (g128, N // 8), dtype=torch.int32, device=DEV)
    sc = (torch.rand(g128, N, device=DEV) * 0.004 + 0.002).to(torch.float16)
    mq, ms, mz = ops.awq_marlin_repack(qw, qz, sc)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain = ops.awq_marlin_gemm(a, mq, ms, mz, M, N, K)
    g = ops.awq_marlin_gemm_deferred(a, mq, ms, mz, M, N, K)
    res0 = torch.randn(M, N, dtype=torch.float16, device=DEV)
    w = (torch.rand(N, device=DEV) + 0.5).half()
    res_a, res_b = res0.clone(), res0.clone()
    ops.fused_add_rms_norm(plain, res_a, w, 1e-5)
    fused = ops.fused_add_rms_norm_splitk(g, res_b, w, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(_bits(fused), _bits(plain)) and torch.equal(_bits(res_a), _bits(res_b))


@pytest.mark.parametrize("M", [1, 16, 64, 256])
def test_deferred_marlin_24_gemm(ops, M):
    """2:4-sparse Marlin GEMM: deferred split-K + fused consumer is bit-identical to the plain sequence."""
    from oracle import packing
    K, N = 4096, 4096
    seed_all(M)
    w = torch.randn(K, N, dtype=torch.float16)
    _, q24, meta, s24 = packing.marlin_24_quantize(w, 4, 128)
    q24, meta, s24 = q24.to(DEV), meta.to(DEV), s24.to(DEV)
    ws = torch.zeros(N // 128 * 64, dtype=torch.int32, device=DEV)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    plain = ops.gptq_marlin_24_gemm(a, q24, meta, s24, ws, 4, M, N, K)
    g = ops.gptq_marlin_24_gemm_deferred(a, q24, meta, s24, ws, 4, M, N, K)
    res0 = torch.randn(M, N, dtype=torch.float16, device=DEV)
    wn = (torch.rand(N, device=DEV) + 0.5).half()
    res_a, res_b = res0.clone(), res0.clone()
    ops.fused_add_rms_norm(plain, res_a, wn, 1e-5)
    fused = ops.fused_add_rms_norm_splitk(g, res_b, wn, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(_bits(fused), _bits(plain)) and torch.equal(_bits(res_a), _bits(res_b))


@pytest.mark.parametrize("M", [1, 16, 64, 256])
@pytest.mark.parametrize("K,N", [(4096, 4096), (14336, 4096), (4096, 6144)])
def test_deferred_fp8_scaled_mm_consumers(ops, M, K, N):
    """fp8 cutlass_scaled_mm with the reduce + scale epilogue deferred: every fused consumer equals the plain sequence bit
    for bit (sum in the reduce kernel's order, sa * (sb * sum) as its epilogue, one rounding), maxima included."""
    seed_all(M + N)
    a = (torch.randn(M, K, device=DEV)).to(torch.float8_e4m3fn)
    b = (torch.randn(N, K, device=DEV)).to(torch.float8_e4m3fn)
    sa = torch.tensor([0.013], device=DEV)
    sb = torch.tensor([0.021], device=DEV)
    plain = ops.cutlass_scaled_mm(a, b.t(), sa, sb, torch.float16)
    g = ops.cutlass_scaled_mm_deferred(a, b.t(), sa, sb, torch.float16)
    res0 = torch.randn(M, N, dtype=torch.float16, device=DEV)
    w = (torch.rand(N, device=DEV) + 0.5).half()
    res_a, res_b = res0.clone(), res0.clone()
    ops.fused_add_rms_norm(plain, res_a, w, 1e-5)
    fused, amax = ops.fused_add_rms_norm_splitk(g, res_b, w, 1e-5, want_absmax=True)
    torch.cuda.synchronize()
    assert torch.equal(_bits(fused), _bits(plain)) and torch.equal(_bits(res_a), _bits(res_b))
    assert torch.equal(amax, plain.float().abs().amax(dim=1))
    # gated activation consumer on a fresh deferred result
    plain2 = ops.cutlass_scaled_mm(a, b.t(), sa, sb, torch.float16)
    g2 = ops.cutlass_scaled_mm_deferred(a, b.t(), sa, sb, torch.float16)
    out_a = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
    out_b = torch.empty_like(out_a)
    ops.silu_and_mul(out_a, plain2)
    amax2 = ops.silu_and_mul_splitk(out_b, g2, want_absmax=True)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out_a), _bits(out_b))
    assert torch.equal(amax2, out_a.float().abs().amax(dim=1))


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("producer", ["o_proj", "down_proj"])
@pytest.mark.parametrize("consumer", ["qkv", "gate_up_act", "o_like"])
@pytest.mark.parametrize("rows", [None, "4"])
def test_norm_fused_gemm(ops, tune, M, producer, consumer, rows):
    """Round 3 (late): fused_add_rms_norm + gptq_marlin_gemm[_silu_and_mul] as ONE launch at batch <= 4
    (fused_add_rms_norm_gptq_marlin_gemm -> nmx_gptq_marlin_gemm_norm: the norm runs in the GEMM's prologue). Producers: the
    deferred K-split slabs of o_proj (4096 -> 4096) and down_proj (14336 -> 4096); consumers: qkv (deferred slabs again),
    gate_up + silu_and_mul, and a 4096 x 4096 matrix. Bit-identical to fused_add_rms_norm_splitk followed by the GEMM op -
    output, residual and the deferred slabs after materialize() -, the oracle's unfused norm on the reduced producer output
    bounds the A operand. By default ONE row is served (more rows cost more than the launch saves); NMX_GEMM_NORM_ROWS=4
    opens the multi-row prologue; rows beyond the limit take the two-launch route through the same op (in-place residual)."""
    tune(NMX_GEMM_NORM_ROWS=rows)
    seed_all(M)
    Kp = 4096 if producer == "o_proj" else 14336
    H = 4096
    qp, sp = _weights(Kp, H, 3)
    a = torch.randn(M, Kp, dtype=torch.float16, device=DEV)
    N = {"qkv": 6144, "gate_up_act": 28672, "o_like": 4096}[consumer]
    act = consumer == "gate_up_act"
    qc, sc = _weights(H, N, 4)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    wsp = torch.zeros(max(N, H) // 64 * 16, dtype=torch.int32, device=DEV)
    res0 = torch.randn(M, H, dtype=torch.float16, device=DEV)
    w = (torch.rand(H, device=DEV) + 0.5).half()

    # the unfused sequence
    g1 = ops.gptq_marlin_gemm_deferred(a, qp, sp, e, e, wsp, 4, M, H, Kp, True)
    assert g1.splits > 1
    res_a = res0.clone()
    h = ops.fused_add_rms_norm_splitk(g1, res_a, w, 1e-5).clone()
    if act:
        want = ops.gptq_marlin_gemm_silu_and_mul(h, qc, sc, e, e, wsp, 4, M, N, H, True)
    else:
        want = ops.gptq_marlin_gemm(h, qc, sc, e, e, wsp, 4, M, N, H, True)

    # the fused op on a fresh deferred producer
    g2 = ops.gptq_marlin_gemm_deferred(a, qp, sp, e, e, wsp, 4, M, H, Kp, True)
    res_b = res0.clone()
    got, res_new = ops.fused_add_rms_norm_gptq_marlin_gemm(g2, res_b, w, 1e-5, qc, sc, e, e, wsp, 4, M, N, H, True,
                                                           silu_and_mul=act)
    torch.cuda.synchronize()
    from neuralmagic_vllm_amd import _lib
    served = bool(_lib.lib().nmx_gptq_marlin_gemm_norm_supported(M, N, H, H // 128, 4, 1, int(act)))
    assert served == (M <= (4 if rows else 1))
    if served:
        assert res_new.data_ptr() != res_b.data_ptr() and torch.equal(_bits(res_b), _bits(res0))  # the input residual is only read
    else:
        assert res_new.data_ptr() == res_b.data_ptr()
    assert torch.equal(_bits(res_new), _bits(res_a))
    out = got if act else got.materialize()
    assert torch.equal(_bits(out), _bits(want))
    assert g2.splits == 1  # consumed
    # the oracle's unfused norm on the reduced producer output, then its GEMM slice on a few columns
    x_o, r_o = ops.gptq_marlin_gemm(a, qp, sp, e, e, wsp, 4, M, H, Kp, True).cpu(), res0.cpu()
    oracle.fused_add_rms_norm(x_o, r_o, w.cpu(), 1e-5)
    assert torch.equal(_bits(res_new.cpu()), _bits(r_o))
    torch.testing.assert_close(h.cpu().float(), x_o.float(), atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("num_seqs", [1, 3, 8, 16, 17])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("rows", [None, "4", "16"])
def test_attn_reduce_gemm(ops, tune, num_seqs, dtype, kv_dtype, rows):
    """Round 3 (late): paged_attention_v2's reduce inside o_proj's prologue (paged_attention_v2_partials ->
    paged_attention_gptq_marlin_gemm -> nmx_gptq_marlin_gemm_attn: every wave of marlin_decode_kernel<ATTN> reduces the heads of
    its own K slice). Llama-3-8B geometry (32 / 8 heads x 128, o_proj 4096 x 4096), ragged sequence lengths, fp16 / bf16, fp16 and
    fp8 KV. The result after materialize() is bit-identical to paged_attention_v2 + gptq_marlin_gemm; the partition results
    reduced on their own (AttnPartials.materialize) equal paged_attention_v2's output bit for bit. The fused form is OFF by default (level
    with the reduce launch); NMX_GEMM_ATTN=rows switches it on up to the kernel's 16; everything else takes reduce launch + GEMM
    through the same op."""
    import random
    tune(NMX_GEMM_ATTN=rows)
    from util import create_kv_caches_with_random
    seed_all(num_seqs)
    nq, nkv, D, block_size = 32, 8, 128, 16
    scale = float(D**-0.5)
    q = torch.empty(num_seqs, nq, D, dtype=dtype, device=DEV).uniform_(-scale, scale)
    seq_lens = [1024, 129, 577, 1000, 1, 512, 513, 64][:num_seqs] + [random.randint(1, 1024) for _ in range(max(0, num_seqs - 8))]
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    bt = torch.tensor([[random.randint(0, 255) for _ in range(mb)] for _ in range(num_seqs)], dtype=torch.int32, device=DEV)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    kcs, vcs = create_kv_caches_with_random(256, block_size, 1, nkv, D, kv_dtype, dtype)
    kc, vc = kcs[0].to(DEV), vcs[0].to(DEV)
    kv_scale = 0.75 if kv_dtype != "auto" else 1.0
    H = nq * D
    qw, sw = _weights(H, H, 9)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    wsp = torch.zeros(H // 64 * 16, dtype=torch.int32, device=DEV)

    # the two-op sequence
    P = (max_len + 511) // 512
    tmp = torch.empty(num_seqs, nq, P, D, dtype=dtype, device=DEV)
    es = torch.empty(num_seqs, nq, P, dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    a = torch.empty_like(q)
    ops.paged_attention_v2(a, es, ml, tmp, q, kc, vc, nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    want = ops.gptq_marlin_gemm(a.view(num_seqs, H), qw, sw.to(dtype), e, e, wsp, 4, num_seqs, H, H, True)

    parts = ops.paged_attention_v2_partials(q, kc, vc, nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    got = ops.paged_attention_gptq_marlin_gemm(parts, qw, sw.to(dtype), e, e, wsp, 4, num_seqs, H, H, True)
    from neuralmagic_vllm_amd import _lib
    served = bool(_lib.lib().nmx_gptq_marlin_gemm_attn_supported(num_seqs, H, H, H // 128, 4, 1 if dtype == torch.float16 else 2, nq, D,
                                                                 parts.tmp.shape[2]))
    assert served == (num_seqs <= (int(rows) if rows else 0))
    assert (parts.out is None) == served  # fused: the attention output never existed as a tensor
    torch.cuda.synchronize()
    assert torch.equal(_bits(got.materialize()), _bits(want))
    parts2 = ops.paged_attention_v2_partials(q, kc, vc, nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    assert torch.equal(_bits(parts2.materialize()), _bits(a))
    # switched off: the same op, two launches, the same bits
    tune(NMX_GEMM_ATTN="0")
    parts3 = ops.paged_attention_v2_partials(q, kc, vc, nkv, scale, bt, sl, block_size, max_len, None, kv_dtype, kv_scale)
    got3 = ops.paged_attention_gptq_marlin_gemm(parts3, qw, sw.to(dtype), e, e, wsp, 4, num_seqs, H, H, True)
    assert parts3.out is not None and torch.equal(_bits(got3.materialize()), _bits(want))


@pytest.mark.parametrize("consumer", ["qkv", "gate_up_act"])
@pytest.mark.parametrize("kind", ["channelwise", "not_ws", "bf16"])
def test_norm_fused_gemm_variants(ops, tune, consumer, kind):
    """The other instantiations of marlin_decode_kernel<NORM> at one row: channel-wise scales (one scale row: GROUPED = false) and
    group scales applied to the fp32 group accumulators instead of the weights (NMX_GEMM_LEAN ws = 0: WS = false); bit-identical to
    the two-launch sequence under the same configuration; and bfloat16 activations / scales / norm weight."""
    seed_all(11)
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    M, H, Kp = 1, 4096, 4096
    qp, sp = _weights(Kp, H, 3)
    sp = sp.to(dt)
    a = torch.randn(M, Kp, dtype=dt, device=DEV)
    N = {"qkv": 6144, "gate_up_act": 28672}[consumer]
    act = consumer == "gate_up_act"
    qc, sc = _weights(H, N, 4)
    sc = sc.to(dt)
    if kind == "channelwise":
        sc = sc[:1].contiguous()
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    wsp = torch.zeros(max(N, H) // 64 * 16, dtype=torch.int32, device=DEV)
    res0 = torch.randn(M, H, dtype=dt, device=DEV)
    w = (torch.rand(H, device=DEV) + 0.5).to(dt)
    g1 = ops.gptq_marlin_gemm_deferred(a, qp, sp, e, e, wsp, 4, M, H, Kp, True)  # the producers under the default dispatch
    g2 = ops.gptq_marlin_gemm_deferred(a, qp, sp, e, e, wsp, 4, M, H, Kp, True)
    assert g1.splits > 1 and g2.splits > 1
    if kind == "not_ws":
        tune(NMX_GEMM_LEAN="4,2,1,0" if not act else "4,1,1,0")
    res_a = res0.clone()
    h = ops.fused_add_rms_norm_splitk(g1, res_a, w, 1e-5).clone()
    want = (ops.gptq_marlin_gemm_silu_and_mul if act else ops.gptq_marlin_gemm)(h, qc, sc, e, e, wsp, 4, M, N, H, True)
    got, res_new = ops.fused_add_rms_norm_gptq_marlin_gemm(g2, res0.clone(), w, 1e-5, qc, sc, e, e, wsp, 4, M, N, H, True, silu_and_mul=act)
    torch.cuda.synchronize()
    assert res_new.data_ptr() != res_a.data_ptr() and torch.equal(_bits(res_new), _bits(res_a))
    out = got if act else got.materialize()
    assert torch.equal(_bits(out), _bits(want))
