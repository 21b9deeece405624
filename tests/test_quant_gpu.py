"""GPU parity tests: fp8 / int8 activation quantisation (codes must equal the oracle / torch exactly) and the W8A8
scaled GEMM. Mirrors tests/quantization/test_fp8.py:57-97, tests/kernels/test_int8_quant.py:24-71 and
tests/kernels/test_cutlass.py:35-114 of the reference."""
import pytest
import torch

import oracle
from util import seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("shape", [(1, 17), (13, 4096), (83, 5120), (512, 1031)])
def test_scaled_fp8_quant(ops, dtype, shape):
    seed_all(0)
    x = (torch.randn(shape) * 13).to(dtype)
    # dynamic
    q, s = ops.scaled_fp8_quant(x.to(DEV))
    amax = x.float().abs().max()
    assert float(s) == float(amax / 448.0)
    expect = (x.float() * (1.0 / s.cpu())).clamp(-448, 448).to(torch.float8_e4m3fn)  # test_fp8.py:61-69
    assert torch.equal(q.cpu().view(torch.uint8), expect.view(torch.uint8))
    qo, so = oracle.scaled_fp8_quant(x)
    assert float(so) == float(s) and torch.equal(q.cpu().view(torch.uint8), qo.view(torch.uint8))
    # static
    st = torch.tensor([0.37], dtype=torch.float32)
    q2, _ = ops.scaled_fp8_quant(x.to(DEV), st.to(DEV))
    qo2, _ = oracle.scaled_fp8_quant(x, st)
    assert torch.equal(q2.cpu().view(torch.uint8), qo2.view(torch.uint8))
    # padding: extra rows allocated, leading rows identical
    q3, _ = ops.scaled_fp8_quant(x.to(DEV), st.to(DEV), batch_dim_padding=shape[0] + 17)
    assert q3.shape[0] == shape[0] + 17
    assert torch.equal(q3[:shape[0]].cpu().view(torch.uint8), qo2.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("num_tokens,hidden", [(1, 16), (7, 67), (83, 5120), (512, 8192)])
def test_scaled_int8_quant(ops, dtype, num_tokens, hidden):
    seed_all(1)
    x = (torch.rand(num_tokens, hidden) * 1000 - 300).to(dtype)
    q, s = ops.scaled_int8_quant(x.to(DEV))
    qo, so = oracle.scaled_int8_quant(x)
    assert torch.equal(s.cpu(), so)
    # rounding of x * (127 / absmax): device and host multiply identically; allow 1 code like the reference (atol=1)
    assert (q.cpu().int() - qo.int()).abs().max() <= 1
    assert (q.cpu() != qo).float().mean() < 1e-3
    st = torch.tensor([2.1], dtype=torch.float32)
    q2, _ = ops.scaled_int8_quant(x.to(DEV), st.to(DEV))
    qo2, _ = oracle.scaled_int8_quant(x, st)
    assert torch.equal(q2.cpu(), qo2)


def to_fp8(t):
    return torch.round(t.clamp(min=-448, max=448)).to(dtype=torch.float8_e4m3fn)


def to_int8(t):
    return torch.round(t.clamp(min=-128, max=127)).to(dtype=torch.int8)


@pytest.mark.parametrize("m,n,k", [(1, 16, 16), (1, 4096, 4096), (16, 6144, 4096), (33, 256, 496), (64, 1024, 128), (83, 512, 1024), (512, 512, 512),
                                   (100, 80, 14336), (64, 4096, 14336), (5, 144, 256), (17, 64, 64)])
@pytest.mark.parametrize("per_act_token", [True, False])
@pytest.mark.parametrize("per_out_ch", [True, False])
@pytest.mark.parametrize("is_fp8", [True, False])
@pytest.mark.parametrize("use_bias", [False, True])
def test_cutlass_scaled_mm(ops, m, n, k, per_act_token, per_out_ch, is_fp8, use_bias):
    """tests/kernels/test_cutlass.py:50-114 (same data recipe)."""
    seed_all(2)
    out_dtype = torch.bfloat16 if is_fp8 else torch.float16
    if is_fp8:
        a = to_fp8(torch.randn(m, k))
        b = to_fp8(torch.randn(n, k).t())
    else:
        a = to_int8(torch.randn(m, k) * 5)
        b = to_int8(torch.randn(n, k).t() * 5)
    sa = torch.randn((m, 1) if per_act_token else (1, 1), dtype=torch.float32) / 10
    sb = torch.randn((1, n) if per_out_ch else (1, 1), dtype=torch.float32) / 10
    bias = (torch.rand(n) * 10).to(out_dtype) if use_bias else None
    bg = b.t().contiguous().to(DEV).t()  # column-major on the device
    out = ops.cutlass_scaled_mm(a.to(DEV), bg, sa.to(DEV), sb.to(DEV), out_dtype, bias.to(DEV) if use_bias else None)
    base = (sa * (sb * torch.mm(a.float(), b.float()))).to(out_dtype)  # baseline_scaled_mm, test_cutlass.py:35-47
    if use_bias:
        base = base + bias
    orc = oracle.scaled_mm(a, b, sa, sb, out_dtype, bias)
    if is_fp8:
        torch.testing.assert_close(out.cpu(), base, rtol=1e-2, atol=5e-2)
    else:
        torch.testing.assert_close(out.cpu(), base, rtol=1e-1, atol=1e0)
    # against the oracle: same math, only the fp32 summation order differs
    torch.testing.assert_close(out.cpu().float(), orc.float(), rtol=1e-2, atol=2e-2 if is_fp8 else 1e-1)


@pytest.mark.parametrize("m,n,k", [(256, 16384, 1024), (129, 16400, 256), (300, 1104, 2048), (256, 6144, 4096), (65, 64, 128)])
@pytest.mark.parametrize("is_fp8", [True, False])
@pytest.mark.parametrize("tile", [None, "0", "2,0,1,4", "2,2,1,4"])
def test_scaled_mm_tile_kernel(ops, tune, m, n, k, is_fp8, tile):
    """M > 64: scaled_mm_tile_kernel (128 x 256 tiles when they fill the chip, else 128 x 128 with K splits; fp8: the forced
    TALL 256 x 128 tile, alone and with two K splits), ragged M / N,
    per-row and per-column scales with bias; checked against the fp32 product computed on the device and against the
    per-wave kernel (NMX_MM_TILE=0) through the same bar. int8 accumulates exactly: the two kernels must agree bit for bit."""
    if tile is not None and tile.endswith(",4") and not is_fp8:
        pytest.skip("the tall tile is an fp8 form")
    seed_all(5)
    out_dtype = torch.bfloat16 if is_fp8 else torch.float16
    if is_fp8:
        a = to_fp8(torch.randn(m, k)).to(DEV)
        b = to_fp8(torch.randn(n, k)).to(DEV)
    else:
        a = to_int8(torch.randn(m, k) * 5).to(DEV)
        b = to_int8(torch.randn(n, k) * 5).to(DEV)
    sa = (torch.rand(m, 1) / 10 + 0.01).to(DEV)
    sb = (torch.rand(1, n) / 10 + 0.01).to(DEV)
    bias = (torch.rand(n) * 10).to(out_dtype).to(DEV)
    tune(NMX_MM_TILE=tile)
    out = ops.cutlass_scaled_mm(a, b.t(), sa, sb, out_dtype, bias)
    base = (sa * (sb * torch.mm(a.float(), b.float().t()))).to(out_dtype) + bias
    torch.testing.assert_close(out, base, rtol=1e-2, atol=5e-2 if is_fp8 else 1e-1)
    if not is_fp8 and tile is None:
        tune(NMX_MM_TILE="0")
        assert torch.equal(out, ops.cutlass_scaled_mm(a, b.t(), sa, sb, out_dtype, bias))


@pytest.mark.parametrize("m", [64, 256, 512])
@pytest.mark.parametrize("k,n", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
def test_scaled_mm_llama3_8b_shapes(ops, k, n, m):
    """configs[3] (Llama-3-8B fp8 W8A8) at its own workload: the four (K, N) `bench.py --config fp8` times, default dispatch
    (the 128 x 256 LDS-DMA tile kernel on gate_up at M = 256), per-tensor scales like Fp8LinearMethod. Checked against the
    fp32 product on the device (test_cutlass.py:35-47's baseline and bars) on every column, against the CPU oracle on a
    128-column slice from both ends of N, and the deferred form + materialize against the plain op bit for bit."""
    g = torch.Generator(device=DEV)
    g.manual_seed(m + n + k)
    a = torch.randn(m, k, device=DEV, generator=g).to(torch.float8_e4m3fn)
    b = torch.randn(n, k, device=DEV, generator=g).to(torch.float8_e4m3fn)
    sa = torch.full((1, 1), 0.037, device=DEV)
    sb = torch.full((1, 1), 0.011, device=DEV)
    out = ops.cutlass_scaled_mm(a, b.t(), sa, sb, torch.bfloat16)
    base = (sa * (sb * torch.mm(a.float(), b.float().t()))).to(torch.bfloat16)
    torch.testing.assert_close(out, base, rtol=1e-2, atol=5e-2)
    for lo in (0, n - 128):
        orc = oracle.scaled_mm(a.cpu(), b[lo:lo + 128].cpu().t(), sa.cpu(), sb.cpu(), torch.bfloat16, None)
        torch.testing.assert_close(out[:, lo:lo + 128].cpu().float(), orc.float(), rtol=1e-2, atol=2e-2)
    d = ops.cutlass_scaled_mm_deferred(a, b.t(), sa, sb, torch.bfloat16)
    assert torch.equal(d.materialize().view(torch.int16), out.view(torch.int16))  # nmx_splitk_reduce_scaled: the GEMM's own reduce


def test_scaled_mm_errors(ops):
    a = torch.zeros(4, 32, dtype=torch.int8, device=DEV)
    b = torch.zeros(32, 16, dtype=torch.int8, device=DEV)  # row-major: must be rejected
    s = torch.ones(1, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="column-major"):
        ops.cutlass_scaled_mm(a, b, s, s, torch.float16)
    assert ops.cutlass_scaled_mm_supports_fp8(95) is True
