"""GPU parity tests for the Marlin family (gptq_marlin_repack bit-exact; gptq_marlin_gemm / marlin_gemm /
fp8_marlin_gemm vs the CPU oracle). Mirrors tests/kernels/test_marlin_gemm.py of the reference: same shape grid,
its bar is mean|d|/mean|ref| < 0.04; ours is <= 1e-3 (north-star), against a.float() @ w_ref.float()."""
import functools

import numpy as np
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, from_bits, load_golden, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

MARLIN_K_CHUNKS = [128]
MARLIN_N_CHUNKS = [64, 128, 256]
MNK_FACTORS = [(1, 1, 1), (1, 4, 8), (1, 7, 5), (13, 17, 67), (26, 37, 13), (67, 13, 11)]
GROUP_SIZES = [-1, 32, 64, 128]
NUM_BITS = [4, 8]
TOL = 1e-3


def workspace(size_n):
    return torch.zeros(size_n // 64 * 16, dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("name", ["marlin_b4_g128_act0", "marlin_b4_g64_act1", "marlin_b8_g-1_act0", "marlin_b8_g32_act1"])
def test_repack_and_gemm_golden(ops, name):
    """Inputs / expected outputs made by the reference's own Python utilities."""
    g = load_golden(name)
    bits = int(g["bits"])
    K, N = g["q_w"].shape
    act = g["g_idx"].size > 0
    sort_idx = torch.from_numpy(g["sort_idx"]).to(DEV) if act else torch.empty(0, dtype=torch.int32, device=DEV)
    out = ops.gptq_marlin_repack(torch.from_numpy(g["q_gptq"]).to(DEV), sort_idx, K, N, bits)
    assert np.array_equal(out.cpu().numpy(), g["marlin_q"])
    g_idx = torch.from_numpy(g["g_idx_sorted"]).to(DEV) if act else torch.empty(0, dtype=torch.int32, device=DEV)
    a = from_bits(g["a"], torch.float16).to(DEV)
    ws = workspace(N)
    c = ops.gptq_marlin_gemm(a, out, from_bits(g["marlin_s"], torch.float16).to(DEV), g_idx, sort_idx, ws, bits,
                             a.shape[0], N, K, True)
    assert compute_max_diff(c.cpu(), torch.from_numpy(g["c_ref"])) < TOL
    assert int(ws.abs().sum()) == 0  # workspace left zeroed


@pytest.mark.parametrize("k_chunk", MARLIN_K_CHUNKS)
@pytest.mark.parametrize("n_chunk", MARLIN_N_CHUNKS)
@pytest.mark.parametrize("num_bits", NUM_BITS)
@pytest.mark.parametrize("group_size", GROUP_SIZES)
@pytest.mark.parametrize("act_order", [False, True])
@pytest.mark.parametrize("mnk_factors", MNK_FACTORS)
def test_marlin_repack(ops, k_chunk, n_chunk, num_bits, group_size, act_order, mnk_factors):
    """tests/kernels/test_marlin_gemm.py:62-114 — exact."""
    _, n_factor, k_factor = mnk_factors
    size_k, size_n = k_chunk * k_factor, n_chunk * n_factor
    if act_order and (group_size == -1 or group_size == size_k):
        pytest.skip("act_order needs groups")
    seed_all(0)
    gs = size_k if group_size == -1 else group_size
    w = torch.randn(size_k, size_n, dtype=torch.float16)
    _, q_w, _, g_idx, _ = packing.quantize_weights(w, num_bits, gs, act_order)
    q_gptq = packing.gptq_pack(q_w, num_bits, size_k, size_n)
    sort_idx = torch.empty(0, dtype=torch.int32)
    if act_order:
        q_w, g_idx, sort_idx = packing.sort_weights(q_w, g_idx)
    expect = packing.marlin_weights(q_w, size_k, size_n, num_bits)
    got = ops.gptq_marlin_repack(q_gptq.to(DEV), sort_idx.to(DEV), size_k, size_n, num_bits)
    assert torch.equal(got.cpu(), expect)


@functools.lru_cache(maxsize=4)
def _quantized_case(size_m, size_k, size_n, num_bits, group_size, act_order):
    """Seeded inputs + the reference quantizer's output (CPU, the slow part of the grid test): the is_k_full = False / True
    cases of one shape run back to back and share it."""
    seed_all(0)
    a = torch.randn(size_m, size_k, dtype=torch.float16)
    w = torch.randn(size_k, size_n, dtype=torch.float16)
    w_ref, mq, ms, g_idx, sort_idx, _ = packing.marlin_quantize(w, num_bits, group_size, act_order)
    return a, w_ref, mq, ms, g_idx, sort_idx


@pytest.mark.parametrize("is_k_full", [False, True])
@pytest.mark.parametrize("k_chunk", MARLIN_K_CHUNKS)
@pytest.mark.parametrize("n_chunk", MARLIN_N_CHUNKS)
@pytest.mark.parametrize("num_bits", NUM_BITS)
@pytest.mark.parametrize("group_size", GROUP_SIZES)
@pytest.mark.parametrize("mnk_factors", MNK_FACTORS)
@pytest.mark.parametrize("act_order", [False, True])
def test_marlin_gemm(ops, k_chunk, n_chunk, num_bits, group_size, mnk_factors, act_order, is_k_full):
    """tests/kernels/test_marlin_gemm.py:126-179"""
    m_factor, n_factor, k_factor = mnk_factors
    size_m, size_k, size_n = m_factor, k_chunk * k_factor, n_chunk * n_factor
    if act_order and (group_size == -1 or group_size == size_k):
        pytest.skip("act_order needs groups")
    if not act_order and not is_k_full:
        pytest.skip("is_k_full only matters with act_order")
    a, w_ref, mq, ms, g_idx, sort_idx = _quantized_case(size_m, size_k, size_n, num_bits, group_size, act_order)
    out = ops.gptq_marlin_gemm(a.to(DEV), mq.to(DEV), ms.to(DEV), g_idx.to(DEV), sort_idx.to(DEV), workspace(size_n),
                               num_bits, size_m, size_n, size_k, is_k_full)
    ref = torch.matmul(a.float(), w_ref.float())
    assert compute_max_diff(out.cpu(), ref) < TOL


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("num_bits,group_size", [(4, 128), (4, -1), (8, 128), (8, -1)])
@pytest.mark.parametrize("size_m", [1, 16, 17, 33, 64, 100])
def test_marlin_gemm_dtypes_vs_oracle(ops, dtype, num_bits, group_size, size_m):
    seed_all(1)
    size_k, size_n = 512, 384
    a = torch.randn(size_m, size_k, dtype=dtype)
    w = torch.randn(size_k, size_n, dtype=torch.float16)
    _, mq, ms, g_idx, sort_idx, _ = packing.marlin_quantize(w, num_bits, group_size, False)
    ms = ms.to(dtype)
    out = ops.gptq_marlin_gemm(a.to(DEV), mq.to(DEV), ms.to(DEV), g_idx.to(DEV), sort_idx.to(DEV), workspace(size_n),
                               num_bits, size_m, size_n, size_k, True)
    orc = oracle.gptq_marlin_gemm(a, mq, ms, g_idx, sort_idx, None, num_bits, size_m, size_n, size_k, True)
    tol = TOL if dtype == torch.float16 else 4e-3  # bf16 output rounding alone is ~2e-3 relative
    assert compute_max_diff(out.cpu(), orc) < tol


@pytest.mark.parametrize("mnk_factors", MNK_FACTORS)
def test_marlin_gemm_checkpoint_format(ops, mnk_factors):
    """marlin_gemm: weights already Marlin-packed (marlin.py:173-231), groups of 128 or channel-wise, fp16."""
    m_factor, n_factor, k_factor = mnk_factors
    size_m, size_k, size_n = m_factor, 128 * k_factor, 256 * n_factor
    seed_all(2)
    for gs in (128, -1):
        a = torch.randn(size_m, size_k, dtype=torch.float16)
        w = torch.randn(size_k, size_n, dtype=torch.float16)
        w_ref, mq, ms, _, _, _ = packing.marlin_quantize(w, 4, gs, False)
        out = ops.marlin_gemm(a.to(DEV), mq.to(DEV), ms.to(DEV), workspace(size_n), size_m, size_n, size_k)
        assert compute_max_diff(out.cpu(), torch.matmul(a.float(), w_ref.float())) < TOL


@pytest.mark.parametrize("n_chunk", MARLIN_N_CHUNKS)
@pytest.mark.parametrize("mnk_factors", MNK_FACTORS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fp8_marlin_gemm(ops, n_chunk, mnk_factors, dtype):
    """tests/kernels/test_marlin_gemm.py:238-304 — W8A16 with e4m3fn weight bytes."""
    m_factor, n_factor, k_factor = mnk_factors
    size_m, size_k, size_n = m_factor, 128 * k_factor, n_chunk * n_factor
    seed_all(3)
    a = torch.randn(size_m, size_k, dtype=dtype)
    w = torch.randn(size_k, size_n, dtype=dtype)
    fp8_w, w_scale = oracle.scaled_fp8_quant(w)  # dynamic per-tensor (checked separately against torch)
    packed = packing.pack_fp8_to_int32(fp8_w)
    mq = ops.gptq_marlin_repack(packed.to(DEV), torch.empty(0, dtype=torch.int32, device=DEV), size_k, size_n, 8)
    scales = w_scale.repeat(1, size_n).to(dtype)
    ms = packing.marlin_permute_scales(scales, size_k, size_n, -1)
    out = ops.fp8_marlin_gemm(a.to(DEV), mq, ms.to(DEV), workspace(size_n), 8, size_m, size_n, size_k)
    orc = oracle.fp8_marlin_gemm(a, mq.cpu(), ms, None, 8, size_m, size_n, size_k)
    tol = TOL if dtype == torch.float16 else 4e-3
    assert compute_max_diff(out.cpu(), orc) < tol
    assert compute_max_diff(out.cpu(), torch.matmul(a.float(), w.float())) < 0.04  # the reference test's own bar


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("num_bits", [4, 8])
@pytest.mark.parametrize("group_size", [-1, 64, 128])
@pytest.mark.parametrize("shape", [(129, 256, 512), (300, 320, 1024), (512, 1024, 4096), (1000, 64, 192)])
def test_marlin_gemm_large_m(ops, tune, dtype, num_bits, group_size, shape):
    """M > 128 runs the 256 x 256-tile prefill kernel (row / column tiles that are not full, K splits, every mode);
    (1000, 64, 192) has K % 64 != 0 and must take the row-block path."""
    size_m, size_n, size_k = shape
    if group_size > 0 and size_k % group_size:
        pytest.skip("K not a multiple of the group")
    seed_all(11)
    tune(NMX_GEMM_LARGE="1", NMX_GEMM_WIDE="0")  # small shapes: force the tile kernel (the heuristic wants >= 192 tiles)
    w = torch.randn(size_k, size_n, dtype=dtype)
    w_ref, mq, ms, _, _, _ = packing.marlin_quantize(w, num_bits, group_size, False)
    a = torch.randn(size_m, size_k, dtype=dtype)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    out = ops.gptq_marlin_gemm(a.to(DEV), mq.to(DEV), ms.to(DEV), e, e, workspace(size_n), num_bits, size_m, size_n,
                               size_k, True)
    ref = a.float() @ w_ref.float()
    tol = TOL if dtype == torch.float16 else 4e-3  # bf16 output rounding: 2^-9 relative
    assert compute_max_diff(out.cpu(), ref) < tol


def test_fp8_marlin_gemm_large_m(ops, tune):
    seed_all(12)
    tune(NMX_GEMM_LARGE="1", NMX_GEMM_WIDE="0")
    size_m, size_n, size_k = 384, 512, 1024
    w = torch.randn(size_k, size_n, dtype=torch.float16)
    w8 = w.to(torch.float8_e4m3fn)
    packed = packing.pack_fp8_to_int32(w8)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    mq = ops.gptq_marlin_repack(packed.to(DEV), e, size_k, size_n, 8)
    scales = packing.marlin_permute_scales(torch.full((1, size_n), 0.5, dtype=torch.float16), size_k, size_n, -1)
    a = torch.randn(size_m, size_k, dtype=torch.float16)
    out = ops.fp8_marlin_gemm(a.to(DEV), mq, scales.to(DEV), workspace(size_n), 8, size_m, size_n, size_k)
    ref = a.float() @ (w8.float() * 0.5)
    assert compute_max_diff(out.cpu(), ref) < TOL


@pytest.mark.parametrize("shape", [(5, 4096, 14336), (64, 4096, 4096), (300, 6144, 4096)])
@pytest.mark.parametrize("scratch_mb", [0, 1])
def test_marlin_gemm_without_or_with_small_scratch(shape, scratch_mb):
    """The C-ABI must stay correct when the caller hands no (or too little) split-K scratch: it degrades to the number
    of K splits that fit instead of allocating (graph capture) or overrunning the buffer."""
    import ctypes
    from neuralmagic_vllm_amd import _lib
    size_m, size_n, size_k = shape
    seed_all(21)
    w = torch.randn(size_k, size_n, dtype=torch.float16) * 0.05
    w_ref, mq, ms, _, _, _ = packing.marlin_quantize(w, 4, 128, False)
    a = torch.randn(size_m, size_k, dtype=torch.float16)
    ag, qg, sg = a.to(DEV), mq.to(DEV), ms.to(DEV)
    c = torch.empty(size_m, size_n, dtype=torch.float16, device=DEV)
    ws = workspace(size_n)
    scratch = torch.empty(scratch_mb << 20, dtype=torch.uint8, device=DEV) if scratch_mb else None
    guard = None
    if scratch is not None:  # canary right behind the scratch buffer
        big = torch.zeros((scratch_mb << 20) + 4096, dtype=torch.uint8, device=DEV)
        scratch, guard = big[:scratch_mb << 20], big[scratch_mb << 20:]
        guard.fill_(0x5a)
    P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    rc = _lib.lib().nmx_gptq_marlin_gemm(P(ag), P(qg), P(sg), P(None), P(None), P(c), ctypes.c_int64(ws.numel()), P(scratch),
                                        ctypes.c_int64(scratch.numel() if scratch is not None else 0), size_m, size_n, size_k,
                                        4, ms.shape[0], 1, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert compute_max_diff(c.cpu(), a.float() @ w_ref.float()) < TOL
    if guard is not None:
        assert bool((guard == 0x5a).all())


def test_marlin_gemm_errors(ops):
    a = torch.zeros(1, 128, dtype=torch.float16, device=DEV)
    mq = torch.zeros(8, 128, dtype=torch.int32, device=DEV)
    s = torch.zeros(1, 64, dtype=torch.float16, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="num_bits must be 4 or 8"):
        ops.gptq_marlin_gemm(a, mq, s, e, e, workspace(64), 3, 1, 64, 128, True)
    with pytest.raises(RuntimeError, match="workspace.numel"):
        ops.gptq_marlin_gemm(a, mq, s, e, e, torch.zeros(1, dtype=torch.int32, device=DEV), 4, 1, 64, 128, True)
    with pytest.raises(RuntimeError, match="Shape mismatch"):
        ops.gptq_marlin_gemm(a, mq, s, e, e, workspace(64), 4, 2, 64, 128, True)


def test_llama3_8b_shapes_property(ops):
    """BASELINE-size shapes (K, N) of Llama-3-8B at M=16: linearity in A and agreement between the direct and a
    column-sliced computation (size-independent properties) + oracle spot check on a column slice."""
    seed_all(5)
    for K, N in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
        M = 16
        mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV)
        ms = (torch.rand(K // 128, N, device=DEV) * 0.01 + 0.005).to(torch.float16)
        a1 = torch.randn(M, K, dtype=torch.float16, device=DEV)
        a2 = torch.randn(M, K, dtype=torch.float16, device=DEV)
        e = torch.empty(0, dtype=torch.int32, device=DEV)
        ws = workspace(N)
        c1 = ops.gptq_marlin_gemm(a1, mq, ms, e, e, ws, 4, M, N, K, True).float()
        c2 = ops.gptq_marlin_gemm(a2, mq, ms, e, e, ws, 4, M, N, K, True).float()
        c12 = ops.gptq_marlin_gemm(((a1.float() + a2.float()) / 2).half(), mq, ms, e, e, ws, 4, M, N, K, True).float()
        assert compute_max_diff(c12, (c1 + c2) / 2) < 3e-3
        # first 128 columns through the oracle (Marlin rows are [K/16, N*2]: a 64-column group is 128 words)
        ncol = 128
        mq_s = mq[:, :ncol * 2].contiguous().cpu()
        ms_s = ms.cpu().reshape(-1, N // 64, 64)[:, :ncol // 64].reshape(-1, ncol).contiguous()
        orc = oracle.gptq_marlin_gemm(a1.cpu(), mq_s, ms_s, None, None, None, 4, M, ncol, K, True)
        assert compute_max_diff(c1[:, :ncol].cpu(), orc) < TOL
