"""GPU parity of every launch shape of marlin_decode_kernel (the barrier-free small-M int4 kernel) and of the 8-wave
128-column tile of marlin_gemm_kernel, forced through the sweep overrides NMX_GEMM_LEAN / NMX_GEMM_CFG, against
a.float() @ w_ref.float() from the oracle's quantizer (bar 1e-3 like the other Marlin tests). Covers what the default
heuristics only reach on the big Llama shapes: 4 / 8 / 16 waves, cross-workgroup K splits with odd unit counts per wave,
both scale placements, channel-wise scales, bf16, ragged M, several row blocks."""
import os

import pytest
import torch

from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


@pytest.fixture
def force(tune):

    def set_(lean=None, cfg=None, wide="0"):
        # the forced marlin_gemm_kernel / decode-kernel shapes are what these tests are about: keep the wide kernel
        # (M > 64) out of the way unless a test asks for the default dispatch
        tune(NMX_GEMM_LEAN=lean, NMX_GEMM_CFG=cfg, NMX_GEMM_WIDE=wide)

    return set_


def make(K, N, group, dtype=torch.float16, seed=0):
    seed_all(seed)
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, marlin_q, marlin_s, _, _, _ = packing.marlin_quantize(w, 4, K if group == -1 else group, False)
    return w_ref.float(), marlin_q.to(DEV), marlin_s.to(dtype).to(DEV)


def run(ops, a, q, s, K, N):
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    return ops.gptq_marlin_gemm(a.to(DEV), q, s, e, e, ws, 4, a.shape[0], N, K, True).float().cpu()


# K = 128 * units: 7 units with 4 waves x 2 splits leaves waves with 1, 0 and odd unit counts
@pytest.mark.parametrize("lean", ["4,1", "4,2", "4,4", "8,1", "8,2", "16,1", "16,2", "4,2,1,0", "8,1,1,0", "8,3"])
@pytest.mark.parametrize("m", [1, 5, 16])
@pytest.mark.parametrize("K,N,group", [(896, 192, 128), (1024, 256, -1), (2048, 64, 256)])
def test_decode_kernel_one_row_tile(ops, force, lean, m, K, N, group):
    w_ref, q, s = make(K, N, group)
    a = torch.randn(m, K, dtype=torch.float16)
    ref = a.float() @ w_ref
    force(lean=lean)
    assert compute_max_diff(run(ops, a, q, s, K, N), ref) < TOL
    force(lean="0")  # same inputs through marlin_gemm_kernel
    assert compute_max_diff(run(ops, a, q, s, K, N), ref) < TOL


@pytest.mark.parametrize("lean", ["4,1,2", "4,2,2", "8,1,2", "8,2,2", "4,2,2,0", "8,1,2,0", "4,1,1", "8,2,1"])
@pytest.mark.parametrize("m", [17, 32, 45, 64])
def test_decode_kernel_row_blocks(ops, force, lean, m):
    """M > 16 MT: several row blocks per column group (gridDim.z), the last one ragged."""
    K, N = 1536, 320
    w_ref, q, s = make(K, N, 128, seed=1)
    a = torch.randn(m, K, dtype=torch.float16)
    force(lean=lean)
    assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL


@pytest.mark.parametrize("lean", ["4,2", "8,1", "16,1", "8,1,2,0", "4,2,2"])
@pytest.mark.parametrize("group", [128, -1])
def test_decode_kernel_bf16(ops, force, lean, group):
    K, N, m = 1024, 192, 19
    w_ref, q, s = make(K, N, group, torch.bfloat16, seed=2)
    a = torch.randn(m, K, dtype=torch.bfloat16)
    force(lean=lean)
    out = run(ops, a, q, s, K, N)
    force(lean="0")
    base = run(ops, a, q, s, K, N)  # marlin_gemm_kernel on the same bf16 inputs (itself pinned by test_marlin_gpu.py)
    assert compute_max_diff(out, base) < 8e-3  # two bf16 roundings of different accumulation orders


@pytest.mark.parametrize("cfg", ["4,2,1,1", "4,2,2,1", "4,4,2,1", "2,2,2", "2,2,4"])
@pytest.mark.parametrize("m", [33, 64, 100])
def test_gemm_kernel_forced_tiles(ops, force, cfg, m):
    """marlin_gemm_kernel tile shapes incl. the 8-wave 128-column one, with the 16-byte weight loads + half exchange."""
    K, N = 2048, 512
    w_ref, q, s = make(K, N, 128, seed=3)
    a = torch.randn(m, K, dtype=torch.float16)
    force(lean="0", cfg=cfg)
    assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL


def test_llama_shapes_default_dispatch(ops, force):
    """The shapes the heuristics were fitted on, default dispatch (decode kernel for qkv / o at small M)."""
    force(wide=None)
    for (K, N) in ((4096, 6144), (4096, 4096)):
        w_ref, q, s = make(K, N, 128, seed=4)
        for m in (1, 16, 32, 64):
            a = torch.randn(m, K, dtype=torch.float16)
            assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL


def test_wide_n_large_m_default_dispatch(ops, force):
    """gate_up-sized N at M = 200 / 256: the 4-wave 256-column workgroups picked when 8-wave ones would leave a ragged
    second round."""
    force(wide=None)
    K, N = 1024, 28672
    w_ref, q, s = make(K, N, 128, seed=5)
    for m in (200, 256):
        a = torch.randn(m, K, dtype=torch.float16)
        assert compute_max_diff(run(ops, a, q, s, K, N), a.float() @ w_ref) < TOL
