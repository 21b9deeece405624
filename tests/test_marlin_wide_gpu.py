"""GPU parity of marlin_wide_kernel (csrc/marlin_wide.hip: the M > 64 path of gptq_marlin_gemm / marlin_gemm /
fp8_marlin_gemm; reference gptq_marlin.cu:1735-1868, test model tests/kernels/test_marlin_gemm.py:126-179).

* every tile shape (WM x WN x WK waves) with and without cross-workgroup K splits, forced through NMX_GEMM_WIDE, on
  ragged M / N / K-slice counts, against a.float() @ w_ref.float() from the oracle's quantizer (bar 1e-3);
* the DEFAULT dispatch on the four real Llama-3-8B (K, N) at M in {128, 256, 512} - the shapes the headline number is
  quoted on - against the CPU oracle on a column slice, plus agreement with the 64-row-block kernel on all columns."""
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3

_cache = {}


def make(K, N, group, bits=4, dtype=torch.float16, seed=0):
    key = (K, N, group, bits, dtype, seed)
    if key not in _cache:
        seed_all(seed)
        w = torch.randn(K, N, dtype=torch.float16)
        w_ref, marlin_q, marlin_s, _, _, _ = packing.marlin_quantize(w, bits, K if group == -1 else group, False)
        _cache[key] = (w_ref.float(), marlin_q.to(DEV), marlin_s.to(dtype).to(DEV))
    return _cache[key]


def run(ops, a, q, s, K, N, bits=4):
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    return ops.gptq_marlin_gemm(a.to(DEV), q, s, e, e, ws, bits, a.shape[0], N, K, True).float().cpu()


# K = 64 * stages: 1536 -> 24 stages (3 splits x 4 slices leaves 2 per slice), 1088 -> 17 stages (ragged slices, some empty)
@pytest.mark.parametrize("wide", ["1,2,1", "1,2,3", "1,4,1", "1,4,2", "2,2,1", "2,2,2", "2,4,1", "2,4,3"])
@pytest.mark.parametrize("m", [65, 128, 200, 256, 300, 513])
@pytest.mark.parametrize("K,N,group", [(1536, 320, 128), (1088, 512, -1), (1024, 192, 64)])
def test_wide_forced_tiles(ops, tune, wide, m, K, N, group):
    w_ref, q, s = make(K, N, group)
    seed_all(m)
    a = torch.randn(m, K, dtype=torch.float16)
    tune(NMX_GEMM_WIDE=wide)
    assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL


@pytest.mark.parametrize("wide", ["1,2,1,4", "1,2,3,4", "1,4,1,4", "1,4,2,4"])
@pytest.mark.parametrize("m", [1, 16, 33, 64])
@pytest.mark.parametrize("K,N,group", [(1536, 320, 128), (1088, 512, -1), (1024, 192, 64)])
def test_wide_64_row_tiles(ops, tune, wide, m, K, N, group):
    """The 64-row wave-tile instantiation (fourth NMX_GEMM_WIDE field = 4; int4, M <= 64)."""
    w_ref, q, s = make(K, N, group)
    seed_all(m)
    a = torch.randn(m, K, dtype=torch.float16)
    tune(NMX_GEMM_WIDE=wide)
    assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL
    w_ref, q, s = make(K, N, group, 4, torch.bfloat16, seed=2)
    ab = torch.randn(m, K, dtype=torch.bfloat16)
    assert compute_max_diff(run(ops, ab, q, s, K, N), ab.float() @ w_ref) < 4e-3


@pytest.mark.parametrize("wide", ["1,2,2", "2,2,1", "2,4,2"])
@pytest.mark.parametrize("bits,group", [(8, 128), (8, -1), (4, 128)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_wide_int8_and_bf16(ops, tune, wide, bits, group, dtype):
    K, N, m = 1024, 384, 270
    w_ref, q, s = make(K, N, group, bits, dtype, seed=1)
    a = torch.randn(m, K, dtype=dtype)
    tune(NMX_GEMM_WIDE=wide)
    out = run(ops, a, q, s, K, N, bits)
    tol = TOL if dtype == torch.float16 else 4e-3  # bf16 output rounding: 2^-9 relative
    assert compute_max_diff(out, a.float() @ w_ref) < tol


@pytest.mark.parametrize("wide", ["1,4,1", "2,2,2"])
def test_wide_fp8(ops, tune, wide):
    seed_all(12)
    size_m, size_n, size_k = 384, 512, 1024
    w8 = torch.randn(size_k, size_n, dtype=torch.float16).to(torch.float8_e4m3fn)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    mq = ops.gptq_marlin_repack(packing.pack_fp8_to_int32(w8).to(DEV), e, size_k, size_n, 8)
    scales = packing.marlin_permute_scales(torch.full((1, size_n), 0.5, dtype=torch.float16), size_k, size_n, -1)
    a = torch.randn(size_m, size_k, dtype=torch.float16)
    ws = torch.zeros(size_n // 64 * 16, dtype=torch.int32, device=DEV)
    tune(NMX_GEMM_WIDE=wide)
    out = ops.fp8_marlin_gemm(a.to(DEV), mq, scales.to(DEV), ws, 8, size_m, size_n, size_k)
    assert compute_max_diff(out.cpu(), a.float() @ (w8.float() * 0.5)) < TOL


@pytest.mark.parametrize("M", [128, 256, 512])
@pytest.mark.parametrize("K,N", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
def test_llama3_8b_shapes_default_dispatch(ops, tune, K, N, M):
    """The dispatch bench.py's headline runs (batch 256) and its neighbours, on the real (K, N): oracle on a column slice
    from both ends of N, and agreement of every column with the 64-row-block kernel (itself pinned by test_marlin_gpu.py)."""
    seed_all(K + N + M)
    mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV)
    ms = (torch.rand(K // 128, N, device=DEV) * 0.01 + 0.005).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    tune(NMX_GEMM_WIDE=None)
    c = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    tune(NMX_GEMM_WIDE="0")
    base = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    assert compute_max_diff(c, base) < TOL
    ncol = 128  # a 64-column group is 128 words of a Marlin row; scales are permuted within 64-column groups
    for lo in (0, N - ncol):
        mq_s = mq[:, lo * 2:(lo + ncol) * 2].contiguous().cpu()
        ms_s = ms[:, lo:lo + ncol].contiguous().cpu()
        orc = oracle.gptq_marlin_gemm(a.cpu(), mq_s, ms_s, None, None, None, 4, M, ncol, K, True)
        assert compute_max_diff(c[:, lo:lo + ncol].cpu(), orc) < TOL


def test_wide_prefill_m2048(ops, tune):
    """M = 2048 (eight row blocks of one column tile on one XCD), checked against the 64-row-block kernel."""
    K, N, M = 4096, 6144, 2048
    seed_all(3)
    mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV)
    ms = (torch.rand(K // 128, N, device=DEV) * 0.01 + 0.005).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    tune(NMX_GEMM_WIDE=None)
    c = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    tune(NMX_GEMM_WIDE="0", NMX_GEMM_LARGE="0")
    base = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    assert compute_max_diff(c, base) < TOL
