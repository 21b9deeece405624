"""GPU parity tests for gptq_marlin_24_gemm (2:4-sparse Marlin on the gfx950 sparse MFMA) vs the CPU oracle.
Mirrors tests/kernels/test_marlin_gemm.py:184-224 of the reference (same shape grid; its bar is 0.04, ours 1e-3)."""
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, from_bits, load_golden, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3

MARLIN_24_K_CHUNKS = [128]
MARLIN_24_N_CHUNKS = [512]
MNK_FACTORS = [(1, 1, 1), (1, 4, 8), (1, 7, 5), (13, 17, 67), (26, 37, 13), (67, 13, 11)]


def workspace24(size_n):
    return torch.zeros(size_n // 128 * 64, dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("name", ["marlin24_b4_g-1", "marlin24_b4_g128", "marlin24_b8_g128"])
def test_marlin_24_golden(ops, name):
    """Packed tensors and expected output produced by the reference's own utilities."""
    g = load_golden(name)
    bits = int(g["bits"])
    a = from_bits(g["a"], torch.float16).to(DEV)
    K, N = g["w"].shape
    ws = workspace24(N)
    c = ops.gptq_marlin_24_gemm(a, torch.from_numpy(g["marlin_24_q"]).to(DEV), torch.from_numpy(g["meta"]).to(DEV),
                                from_bits(g["marlin_24_s"], torch.float16).to(DEV), ws, bits, a.shape[0], N, K)
    assert compute_max_diff(c.cpu(), torch.from_numpy(g["c_ref"])) < TOL
    assert int(ws.abs().sum()) == 0


@pytest.mark.parametrize("k_chunk", MARLIN_24_K_CHUNKS)
@pytest.mark.parametrize("n_chunk", MARLIN_24_N_CHUNKS)
@pytest.mark.parametrize("num_bits", [4, 8])
@pytest.mark.parametrize("group_size", [-1, 128])
@pytest.mark.parametrize("mnk_factors", MNK_FACTORS)
def test_marlin_24_gemm(ops, k_chunk, n_chunk, num_bits, group_size, mnk_factors):
    m_factor, n_factor, k_factor = mnk_factors
    size_m, size_k, size_n = m_factor, k_chunk * k_factor, n_chunk * n_factor
    seed_all(0)
    a = torch.randn(size_m, size_k, dtype=torch.float16)
    w = torch.randn(size_k, size_n, dtype=torch.float16)
    w_24_ref, mq, meta, ms = packing.marlin_24_quantize(w, num_bits, group_size)
    c = ops.gptq_marlin_24_gemm(a.to(DEV), mq.to(DEV), meta.to(DEV), ms.to(DEV), workspace24(size_n), num_bits, size_m,
                                size_n, size_k)
    ref = a.float() @ w_24_ref.float()
    assert compute_max_diff(c.cpu(), ref) < TOL
    # and the oracle's own reading of the packed tensors agrees
    c_or = oracle.gptq_marlin_24_gemm(a, mq, meta, ms, None, num_bits, size_m, size_n, size_k)
    assert compute_max_diff(c.cpu(), c_or) < TOL


@pytest.mark.parametrize("size_m", [1, 16, 17, 33, 64, 100])
@pytest.mark.parametrize("num_bits", [4, 8])
def test_marlin_24_tile_configs(ops, size_m, num_bits):
    """Every (row tiles, column groups) kernel configuration, with and without K splits, K not a multiple of 128."""
    seed_all(1)
    for size_k, size_n, gs in ((192, 128, -1), (1024, 256, 128), (4096, 1024, 128)):
        a = torch.randn(size_m, size_k, dtype=torch.float16)
        w = torch.randn(size_k, size_n, dtype=torch.float16)
        w_24_ref, mq, meta, ms = packing.marlin_24_quantize(w, num_bits, gs)
        c = ops.gptq_marlin_24_gemm(a.to(DEV), mq.to(DEV), meta.to(DEV), ms.to(DEV), workspace24(size_n), num_bits,
                                    size_m, size_n, size_k)
        assert compute_max_diff(c.cpu(), a.float() @ w_24_ref.float()) < TOL, (size_k, size_n, gs)


def random_valid_meta(rows, cols, gen):
    """Random VALID 2:4 metadata: every nibble is (idx0 | idx1 << 2) with idx0 < idx1 (six encodings), four nibbles per int16."""
    nibs = torch.tensor([4, 8, 12, 9, 13, 14], dtype=torch.int32)
    m4 = nibs[torch.randint(0, 6, (rows, cols, 4), generator=gen)]
    m = m4[..., 0] | (m4[..., 1] << 4) | (m4[..., 2] << 8) | (m4[..., 3] << 12)
    return torch.where(m >= 32768, m - 65536, m).to(torch.int16)


@pytest.mark.parametrize("M", [64, 256])
@pytest.mark.parametrize("K,N", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
def test_llama3_8b_shapes_sparse24(ops, K, N, M):
    """configs[2] (Llama-3-8B 2:4-sparse + int4) at ITS OWN workload: the four (K, N) that `bench.py --config sparse24` times,
    default dispatch, against the oracle on a 128-column slice from both ends of N (the CUTLASS metadata reorder and the
    scale permutation are local to 64-column groups, so a column slice of the packed tensors is itself a valid problem).
    Random compressed words + random valid metadata: every kept-position pattern occurs."""
    gen = torch.Generator().manual_seed(K + N + M)
    mq = torch.randint(-2**31, 2**31 - 1, (K // 32, N * 2), dtype=torch.int32, generator=gen)
    meta = random_valid_meta(K // 32, N * 2, gen)
    ms = (torch.rand(K // 128, N, generator=gen) * 0.01 + 0.005).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, generator=gen)
    ws = workspace24(N)
    c = ops.gptq_marlin_24_gemm(a.to(DEV), mq.to(DEV), meta.to(DEV), ms.to(DEV), ws, 4, M, N, K).float().cpu()
    assert int(ws.abs().sum()) == 0
    ncol = 128
    for lo in (0, N - ncol):
        orc = oracle.gptq_marlin_24_gemm(a, mq[:, lo * 2:(lo + ncol) * 2].contiguous(), meta[:, lo * 2:(lo + ncol) * 2].contiguous(),
                                         ms[:, lo:lo + ncol].contiguous(), None, 4, M, ncol, K)
        assert compute_max_diff(c[:, lo:lo + ncol], orc) < TOL, (K, N, M, lo)
    # the deferred form + its consumer-side reduce must give the same bits as the plain op (what the bench's fused step runs)
    d = ops.gptq_marlin_24_gemm_deferred(a.to(DEV), mq.to(DEV), meta.to(DEV), ms.to(DEV), ws, 4, M, N, K)
    assert torch.equal(d.materialize().float().cpu(), c)


@pytest.mark.parametrize("wide", ["1,4,1", "1,4,2", "1,2,1", "1,2,2", "1,2,2,4", None])
@pytest.mark.parametrize("num_bits,group", [(4, -1), (4, 128), (8, -1), (8, 128)])
@pytest.mark.parametrize("M,N,K", [(65, 256, 256), (128, 512, 512), (200, 384, 1024), (256, 1024, 448), (300, 128, 2048)])
def test_marlin_24_wide_kernel(ops, tune, M, N, K, num_bits, group, wide):
    """Round 3: marlin_wide_kernel<SP = true> (M > 64: 128-row / 64-row wave tiles on the sparse MFMA) in every instantiated
    shape - forced through NMX_GEMM_WIDE, K splits included, ragged rows, padding column groups, K / 64 odd - against the
    row-block kernel (NMX_GEMM_WIDE=0) on the same random compressed words + valid metadata, and both against fp32 torch."""
    if group != -1 and K % group != 0:
        pytest.skip("K not a multiple of the group")
    if wide is not None and wide.endswith(",4") and num_bits != 4:
        pytest.skip("64-row wave tiles: int4 only")
    gen = torch.Generator().manual_seed(M + N + K + num_bits)
    pack = 32 // num_bits
    mq = torch.randint(-2**31, 2**31 - 1, (K // 32, N * 16 // pack), dtype=torch.int32, generator=gen).to(DEV)
    meta = random_valid_meta(K // 32, N * 2, gen).to(DEV)
    groups = 1 if group == -1 else K // group
    ms = (torch.rand(groups, N, generator=gen) * 0.01 + 0.005).to(torch.float16).to(DEV)
    a = torch.randn(M, K, dtype=torch.float16, generator=gen).to(DEV)
    ws = workspace24(N)
    tune(NMX_GEMM_WIDE="0")
    base = ops.gptq_marlin_24_gemm(a, mq, meta, ms, ws, num_bits, M, N, K).float()
    tune(NMX_GEMM_WIDE=wide)
    out = ops.gptq_marlin_24_gemm(a, mq, meta, ms, ws, num_bits, M, N, K).float()
    torch.cuda.synchronize()
    assert compute_max_diff(out.cpu(), base.cpu()) < TOL
    d = ops.gptq_marlin_24_gemm_deferred(a, mq, meta, ms, ws, num_bits, M, N, K)
    assert torch.equal(d.materialize().float(), out)


def test_marlin_24_errors(ops):
    a = torch.zeros(1, 128, dtype=torch.float16, device=DEV)
    q = torch.zeros(4, 256, dtype=torch.int32, device=DEV)
    meta = torch.zeros(4, 256, dtype=torch.int16, device=DEV)
    s = torch.ones(1, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="num_bits must be 4 or 8"):
        ops.gptq_marlin_24_gemm(a, q, meta, s, workspace24(128), 3, 1, 128, 128)
    with pytest.raises(RuntimeError, match="below min_workspace_size"):
        ops.gptq_marlin_24_gemm(a, q, meta, s, torch.zeros(1, dtype=torch.int32, device=DEV), 4, 1, 128, 128)
    with pytest.raises(RuntimeError, match="b_meta.size"):
        ops.gptq_marlin_24_gemm(a, q, meta[:2], s, workspace24(128), 4, 1, 128, 128)
    with pytest.raises(RuntimeError, match="float16"):
        ops.gptq_marlin_24_gemm(a.bfloat16(), q, meta, s.bfloat16(), workspace24(128), 4, 1, 128, 128)
    with pytest.raises(RuntimeError, match="Unexpected groupsize"):
        ops.gptq_marlin_24_gemm(torch.zeros(1, 256, dtype=torch.float16, device=DEV),
                                torch.zeros(8, 256, dtype=torch.int32, device=DEV),
                                torch.zeros(8, 256, dtype=torch.int16, device=DEV),
                                torch.ones(4, 128, dtype=torch.float16, device=DEV), workspace24(128), 4, 1, 128, 256)
