"""GPU parity tests of marlin_dma_kernel (csrc/marlin_dma.hip: the fp16 int4 Marlin GEMM with both operands delivered by LDS-DMA),
forced through NMX_GEMM_DMA=<K splits> so that every shape below really runs on it. Reference: the same expectation as
tests/kernels/test_marlin_gemm.py:126-179 (a @ w_ref, w_ref = fp16((q - 8) * s)) with the north star's 1e-3 bar, the CPU
oracle on column slices, and bit-identity of the deferred / fused forms with the plain op."""
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def make(M, N, K, group, seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    q = torch.randint(0, 16, (K, N), dtype=torch.int32, device=DEV, generator=g)
    groups = 1 if group == -1 else K // group
    s = (torch.rand(groups, N, device=DEV, generator=g) * 0.01 + 0.002).to(torch.float16)
    w_ref = ((q - 8).to(torch.float16).view(groups, K // groups, N) * s[:, None, :]).view(K, N)
    shifts = (4 * torch.arange(8, device=DEV, dtype=torch.int32)).view(1, 8, 1)
    packed = (q.view(K // 8, 8, N) << shifts).sum(dim=1, dtype=torch.int32)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV, generator=g)
    return a, packed, s, w_ref


def run(ops, a, packed, s, M, N, K, group):
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(max(N // 64 * 16, 16), dtype=torch.int32, device=DEV)
    mq = ops.gptq_marlin_repack(packed, e, K, N, 4)
    ms = packing.marlin_permute_scales(s, K, N, group)
    return mq, ms, e, ws


@pytest.mark.parametrize("splits", ["1", "2", "4", "1,1", "2,1", "1,2", "2,2", "4,2"])  # "s,1": loader / consumer split; "s,2": marlin_pc_kernel
@pytest.mark.parametrize("group", [128, 64, -1])
@pytest.mark.parametrize("M,N,K", [(128, 256, 256), (70, 320, 512), (130, 512, 448), (256, 1024, 1024), (300, 192, 2048),
                                   (1, 64, 128), (257, 4096, 896)])
def test_dma_kernel_shapes(ops, tune, M, N, K, group, splits):
    """ragged rows (clamped DMA rows, masked stores), column tiles with padding groups (N % 256 != 0), K-groups with unequal
    stage counts (K / 64 odd), 1 / 2 / 4 K splits (fp32 slabs + reduce), grouped (64, 128) and channel-wise scales."""
    if group != -1 and K % group != 0:
        pytest.skip("K not a multiple of the group")
    a, packed, s, w_ref = make(M, N, K, group, M + N + K)
    mq, ms, e, ws = run(ops, a, packed, s, M, N, K, group)
    tune(NMX_GEMM_DMA=str(splits))
    out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    ref = a.float() @ w_ref.float()
    torch.cuda.synchronize()
    assert compute_max_diff(out.float().cpu(), ref.cpu()) < TOL
    # the deferred form: slabs left for the consumer, materialize() = the reduce launch -> the same bits
    d = ops.gptq_marlin_gemm_deferred(a, mq, ms, e, e, ws, 4, M, N, K, True)
    assert torch.equal(d.materialize().view(torch.int16), out.view(torch.int16))
    # against the kernel it replaces (NMX_GEMM_DMA=0): same weights, only the summation order differs
    tune(NMX_GEMM_DMA="0")
    base = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    assert compute_max_diff(out.float().cpu(), base.float().cpu()) < TOL


@pytest.mark.parametrize("mode", ["1", "1,2"])
@pytest.mark.parametrize("group", [128, -1])
@pytest.mark.parametrize("M,N,K", [(128, 512, 256), (200, 1024, 512), (256, 3072, 1024)])
def test_dma_fused_silu_and_mul(ops, tune, M, N, K, group, mode):
    """gate | up column groups in one workgroup + the activation in the epilogue: the same bits as GEMM + silu_and_mul."""
    a, packed, s, w_ref = make(M, N, K, group, 7 * M + N + K)
    mq, ms, e, ws = run(ops, a, packed, s, M, N, K, group)
    tune(NMX_GEMM_DMA=mode)
    out = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True)
    two = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
    ops.silu_and_mul(two, out)
    one = ops.gptq_marlin_gemm_silu_and_mul(a, mq, ms, e, e, ws, 4, M, N, K, True)
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))


@pytest.mark.parametrize("M", [128, 256])
@pytest.mark.parametrize("K,N,splits", [(4096, 6144, "4"), (4096, 4096, "8"), (4096, 28672, "1"), (14336, 4096, "8"),
                                        (4096, 6144, "4,2"), (4096, 28672, "1,2"), (14336, 4096, "8,2")])
def test_dma_llama3_8b_shapes_vs_oracle(ops, tune, K, N, M, splits):
    """The four Llama-3-8B (K, N) on the DMA kernel against the CPU oracle on a 128-column slice from both ends of N."""
    seed_all(K + N + M)
    mq = torch.randint(-2**31, 2**31 - 1, (K // 16, N * 2), dtype=torch.int32, device=DEV)
    ms = (torch.rand(K // 128, N, device=DEV) * 0.01 + 0.005).to(torch.float16)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    tune(NMX_GEMM_DMA=str(splits))
    c = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    ncol = 128
    for lo in (0, N - ncol):
        mq_s = mq[:, lo * 2:(lo + ncol) * 2].contiguous().cpu()
        ms_s = ms[:, lo:lo + ncol].contiguous().cpu()
        orc = oracle.gptq_marlin_gemm(a.cpu(), mq_s, ms_s, None, None, None, 4, M, ncol, K, True)
        assert compute_max_diff(c[:, lo:lo + ncol].cpu(), orc) < TOL
    tune(NMX_GEMM_DMA="0")
    base = ops.gptq_marlin_gemm(a, mq, ms, e, e, ws, 4, M, N, K, True).float()
    assert compute_max_diff(c, base) < TOL
