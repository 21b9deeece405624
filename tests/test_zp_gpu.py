"""GPU parity tests for the zero-point int4 formats: awq_gemm / awq_dequantize / gptq_gemm / gptq_shuffle.
The reference has NO kernel-level test for these ops (parity unpinned); the oracle follows the formulas
of awq/gemm_kernels.cu:367-431 and gptq/q_gemm.cu:1387-1417, and the expected GEMM is a.float() @ w_ref.float()."""
import numpy as np
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


@pytest.mark.parametrize("m", [1, 7, 16, 17, 33, 100])
@pytest.mark.parametrize("k,n", [(128, 128), (512, 384), (1024, 1280), (3584, 8192)])
@pytest.mark.parametrize("group", [32, 128])
def test_awq_gemm(ops, m, k, n, group):
    if (k, n) == (3584, 8192) and m not in (1, 16):
        pytest.skip("large shape covered at two batch sizes")
    seed_all(0)
    w = torch.randn(k, n)
    a = torch.randn(m, k, dtype=torch.float16)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(w, group)
    # both argument orders seen in the reference (awq.py:172 vs _custom_ops.py:175-177)
    out = ops.awq_gemm(a.to(DEV), qweight.to(DEV), scales.to(DEV), qzeros.to(DEV), 8)
    ref = a.float() @ w_ref.float()
    assert compute_max_diff(out.cpu(), ref) < TOL
    out2 = ops.awq_gemm(a.to(DEV), qweight.to(DEV), qzeros.to(DEV), scales.to(DEV), 8)
    assert torch.equal(out, out2)
    orc = oracle.awq_gemm(a, qweight, scales, qzeros, 8)
    assert compute_max_diff(out.cpu(), orc) < TOL


@pytest.mark.parametrize("k,n,group", [(128, 64, 32), (512, 384, 128), (1024, 1280, 64)])
def test_awq_dequantize(ops, k, n, group):
    seed_all(1)
    w = torch.randn(k, n)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(w, group)
    out = ops.awq_dequantize(qweight.to(DEV), scales.to(DEV), qzeros.to(DEV), 0, 0, 0)
    assert torch.equal(out.cpu().view(torch.int16), oracle.awq_dequantize(qweight, scales, qzeros).view(torch.int16))
    assert torch.equal(out.cpu().view(torch.int16), w_ref.view(torch.int16))


def test_awq_errors(ops):
    a = torch.zeros(1, 64, dtype=torch.float16, device=DEV)
    qw = torch.zeros(64, 4, dtype=torch.int32, device=DEV)  # OC = 32: not a multiple of 64
    s = torch.zeros(2, 32, dtype=torch.float16, device=DEV)
    z = torch.zeros(2, 4, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="OC is not multiple"):
        ops.awq_gemm(a, qw, s, z, 8)


def shuffle_ref(qweight: torch.Tensor, q_perm=None) -> torch.Tensor:
    """numpy restatement of gptq_shuffle 4-bit (q_gemm.cu:1543-1553 + make_sequential :1602-1640)."""
    w = qweight.numpy().astype(np.uint32)
    K8, N = w.shape
    nib = np.stack([(w >> (4 * i)) & 0xf for i in range(8)], axis=1).reshape(K8 * 8, N)  # [K, N] codes
    if q_perm is not None:
        nib = nib[q_perm.numpy().astype(np.int64)]
    nib = nib.reshape(K8, 8, N)
    order = [0, 2, 4, 6, 1, 3, 5, 7]  # nibble position p holds k = order[p]
    out = np.zeros((K8, N), dtype=np.uint32)
    for p_, kk in enumerate(order):
        out |= nib[:, kk, :].astype(np.uint32) << (4 * p_)
    return torch.from_numpy(out.astype(np.int32))


@pytest.mark.parametrize("act_order", [False, True])
def test_gptq_shuffle(ops, act_order):
    seed_all(2)
    K, N = 256, 192
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32)
    perm = torch.randperm(K).to(torch.int32) if act_order else torch.empty(0, dtype=torch.int32)
    g = qw.clone().to(DEV)
    ops.gptq_shuffle(g, perm.to(DEV), 4)
    assert torch.equal(g.cpu(), shuffle_ref(qw, perm if act_order else None))


@pytest.mark.parametrize("m", [1, 16, 17, 50, 64])
@pytest.mark.parametrize("k,n", [(128, 64), (512, 384), (4096, 4096)])
@pytest.mark.parametrize("group", [32, 128])
@pytest.mark.parametrize("mode", ["exllama", "exllama_act_order", "plain", "plain_act_order"])
def test_gptq_gemm(ops, m, k, n, group, mode):
    if (k, n) == (4096, 4096) and m not in (1, 16):
        pytest.skip("large shape covered at two batch sizes")
    seed_all(3)
    w = torch.randn(k, n)
    a = torch.randn(m, k, dtype=torch.float16)
    w_ref, qweight, qzeros, scales, g_idx = packing.gptq_quantize(w, 4, group)
    ref = a.float() @ w_ref.float()
    if mode == "exllama":
        qg = qweight.clone().to(DEV)
        ops.gptq_shuffle(qg, torch.empty(0, dtype=torch.int32, device=DEV), 4)
        out = ops.gptq_gemm(a.to(DEV), qg, qzeros.to(DEV), scales.to(DEV), torch.empty(0, device=DEV), True, 4)
    elif mode == "exllama_act_order":
        # checkpoint with desc_act: rows in arbitrary group order; the layer sorts them (gptq.py:212-222)
        perm = torch.randperm(k)
        codes = oracle_codes(qweight)[perm]           # checkpoint row order
        g_idx_ck = g_idx[perm].contiguous()
        qweight_ck = packing.gptq_pack(codes, 4, k, n)
        a_ck = a[:, perm].contiguous()                # activations in checkpoint order: same product
        q_perm = torch.argsort(g_idx_ck).to(torch.int32)
        qg = qweight_ck.to(DEV)
        ops.gptq_shuffle(qg, q_perm.to(DEV), 4)
        out = ops.gptq_gemm(a_ck.to(DEV), qg, qzeros.to(DEV), scales.to(DEV), q_perm.to(DEV), True, 4)
    elif mode == "plain":
        out = ops.gptq_gemm(a.to(DEV), qweight.to(DEV), qzeros.to(DEV), scales.to(DEV), torch.empty(0, device=DEV), False, 4)
    else:
        perm = torch.randperm(k)
        codes = oracle_codes(qweight)[perm]
        g_idx_ck = g_idx[perm].contiguous()
        qweight_ck = packing.gptq_pack(codes, 4, k, n)
        a_ck = a[:, perm].contiguous()
        out = ops.gptq_gemm(a_ck.to(DEV), qweight_ck.to(DEV), qzeros.to(DEV), scales.to(DEV), g_idx_ck.to(DEV), False, 4)
        orc = oracle.gptq_gemm(a_ck, qweight_ck, qzeros, scales, g_idx_ck, 4)
        assert compute_max_diff(out.cpu(), orc) < TOL
    assert compute_max_diff(out.cpu(), ref) < TOL
    if mode == "plain":
        orc = oracle.gptq_gemm(a, qweight, qzeros, scales, None, 4)
        assert compute_max_diff(out.cpu(), orc) < TOL


def oracle_codes(qweight: torch.Tensor) -> torch.Tensor:
    w = qweight.numpy().astype(np.uint32)
    nib = np.stack([(w >> (4 * i)) & 0xf for i in range(8)], axis=1).reshape(w.shape[0] * 8, w.shape[1])
    return torch.from_numpy(nib.astype(np.int32))


@pytest.mark.parametrize("bits", [2, 3, 8])
@pytest.mark.parametrize("m", [1, 17, 40])
@pytest.mark.parametrize("k,n,group", [(128, 64, 32), (512, 384, 128), (2048, 1024, 128)])
@pytest.mark.parametrize("mode", ["exllama", "exllama_act_order", "plain", "plain_act_order"])
def test_gptq_gemm_other_bits(ops, bits, m, k, n, group, mode):
    """2 / 3 / 8-bit GPTQ (q_gemm.cu:329-700; 3-bit = 32 codes per 3 words): both entry conventions, act-order through
    gptq_shuffle's row permutation or through g_idx."""
    seed_all(4)
    w = torch.randn(k, n)
    a = torch.randn(m, k, dtype=torch.float16)
    w_ref, qweight, qzeros, scales, g_idx = packing.gptq_quantize(w, bits, group)
    assert qweight.shape == (k * bits // 32, n) and qzeros.shape == (k // group, n * bits // 32)
    ref = a.float() @ w_ref.float()
    none = torch.empty(0, device=DEV)
    if mode == "exllama":
        qg = qweight.clone().to(DEV)
        ops.gptq_shuffle(qg, torch.empty(0, dtype=torch.int32, device=DEV), bits)
        out = ops.gptq_gemm(a.to(DEV), qg, qzeros.to(DEV), scales.to(DEV), none, True, bits)
    elif mode == "plain":
        out = ops.gptq_gemm(a.to(DEV), qweight.to(DEV), qzeros.to(DEV), scales.to(DEV), none, False, bits)
        assert compute_max_diff(out.cpu(), oracle.gptq_gemm(a, qweight, qzeros, scales, None, bits)) < TOL
    else:
        perm = torch.randperm(k)
        codes = torch.from_numpy(_codes(qweight, bits, k))[perm]  # checkpoint rows in arbitrary group order
        g_idx_ck = g_idx[perm].contiguous()
        qweight_ck = packing.gptq_pack(codes, bits, k, n)
        a_ck = a[:, perm].contiguous()
        if mode == "exllama_act_order":
            q_perm = torch.argsort(g_idx_ck).to(torch.int32)
            qg = qweight_ck.to(DEV)
            ops.gptq_shuffle(qg, q_perm.to(DEV), bits)
            out = ops.gptq_gemm(a_ck.to(DEV), qg, qzeros.to(DEV), scales.to(DEV), q_perm.to(DEV), True, bits)
        else:
            out = ops.gptq_gemm(a_ck.to(DEV), qweight_ck.to(DEV), qzeros.to(DEV), scales.to(DEV), g_idx_ck.to(DEV), False, bits)
            assert compute_max_diff(out.cpu(), oracle.gptq_gemm(a_ck, qweight_ck, qzeros, scales, g_idx_ck, bits)) < TOL
    assert compute_max_diff(out.cpu(), ref) < TOL


def _codes(qweight: torch.Tensor, bits: int, k: int) -> np.ndarray:
    """inverse of packing.gptq_pack: [K * bits / 32, N] words -> [K, N] codes (contiguous bit stream along K)."""
    w = qweight.numpy().view(np.uint32).astype(np.uint64)
    out = np.zeros((k, w.shape[1]), dtype=np.int32)
    for i in range(k):
        pos = i * bits
        v = w[pos // 32] >> (pos % 32)
        if pos % 32 + bits > 32:
            v = v | (w[pos // 32 + 1] << (32 - pos % 32))
        out[i] = (v & ((1 << bits) - 1)).astype(np.int32)
    return out


def test_gptq_bad_bits(ops):
    a = torch.zeros(1, 128, dtype=torch.float16, device=DEV)
    qw = torch.zeros(128 // 4, 64, dtype=torch.int32, device=DEV)
    z = torch.zeros(1, 16, dtype=torch.int32, device=DEV)
    s = torch.zeros(1, 64, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="bit must be 2, 3, 4 or 8"):
        ops.gptq_gemm(a, qw, z, s, torch.empty(0, device=DEV), True, 5)


@pytest.mark.parametrize("m", [1, 16, 48, 200])
@pytest.mark.parametrize("bits,group", [(4, 128), (4, 64), (8, 128)])
def test_cross_format_agreement(ops, m, bits, group):
    """The extra pin available for the zero-point kernels (the reference holds no kernel test for them): ONE symmetric
    fake-quant checkpoint - codes, scales and w_ref from the restatement of the reference's quantize_weights / gptq_pack,
    which tests/test_oracle_golden.py holds bit-exact to the reference's own utilities - served through
      (a) gptq_marlin_repack + gptq_marlin_gemm   (pinned by the reference's test_marlin_gemm.py bar and goldens),
      (b) gptq_gemm, exllama (after gptq_shuffle) and plain entries, zero point 2^(bits-1) stored as z - 1,
      (c) awq_gemm on the same codes in AWQ nibble order (4 bit)
    must give the same product."""
    seed_all(7)
    K, N = 1024, 768
    w = torch.randn(K, N, dtype=torch.float16)
    a = torch.randn(m, K, dtype=torch.float16)
    w_ref, q_w, s, _, _ = packing.quantize_weights(w, bits, group, False)
    ref = a.float() @ w_ref.float()
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    qweight = packing.gptq_pack(q_w, bits, K, N)
    mq = ops.gptq_marlin_repack(qweight.to(DEV), e, K, N, bits)
    ms = packing.marlin_permute_scales(s, K, N, group)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    c_marlin = ops.gptq_marlin_gemm(a.to(DEV), mq, ms.to(DEV), e, e, ws, bits, m, N, K, True).float().cpu()
    assert compute_max_diff(c_marlin, ref) < TOL
    zeros = torch.full((K // group, N), 2**(bits - 1), dtype=torch.int32)
    qzeros = packing.gptq_pack_zeros(zeros, bits)
    noidx = torch.empty(0, device=DEV)
    c_plain = ops.gptq_gemm(a.to(DEV), qweight.to(DEV), qzeros.to(DEV), s.to(DEV), noidx, False, bits).float().cpu()
    qg = qweight.clone().to(DEV)
    ops.gptq_shuffle(qg, e, bits)
    c_exl = ops.gptq_gemm(a.to(DEV), qg, qzeros.to(DEV), s.to(DEV), noidx, True, bits).float().cpu()
    for c in (c_plain, c_exl):
        assert compute_max_diff(c, ref) < TOL
        assert compute_max_diff(c, c_marlin) < TOL
    if bits == 4:
        c_awq = ops.awq_gemm(a.to(DEV), packing.awq_pack(q_w).to(DEV), s.to(DEV), packing.awq_pack(zeros).to(DEV), 8).float().cpu()
        assert compute_max_diff(c_awq, ref) < TOL and compute_max_diff(c_awq, c_marlin) < TOL
        deq = ops.awq_dequantize(packing.awq_pack(q_w).to(DEV), s.to(DEV), packing.awq_pack(zeros).to(DEV), 0, 0, 0).cpu()
        assert torch.equal(deq.view(torch.int16), w_ref.view(torch.int16))  # the reference quantizer's own w_ref, bit for bit


@pytest.mark.parametrize("m", [1, 16, 33, 64, 100, 256, 300])
@pytest.mark.parametrize("k,n", [(1024, 1280), (8192, 1280), (1024, 8192), (3584, 8192), (512, 256)])
def test_awq_marlin_path(ops, m, k, n):
    """AWQ checkpoint -> awq_marlin_repack -> awq_marlin_gemm (what AWQLinearMethod runs after loading): same bar as
    awq_gemm against a @ w_ref, and agreement with the checkpoint-layout op on the same weights. The shapes are the
    Llama-3-70B / TP=8 per-rank ones (BASELINE configs[4])."""
    if (k, n) in ((8192, 1280), (3584, 8192)) and m not in (1, 64, 256):
        pytest.skip("large shapes covered at three batch sizes")
    seed_all(m)
    w = torch.randn(k, n)
    a = torch.randn(m, k, dtype=torch.float16)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(w, 128)
    assert ops.awq_marlin_supported(n, k, k // 128)
    mq, ms, mz = ops.awq_marlin_repack(qweight.to(DEV), qzeros.to(DEV), scales.to(DEV))
    out = ops.awq_marlin_gemm(a.to(DEV), mq, ms, mz, m, n, k).float().cpu()
    assert compute_max_diff(out, a.float() @ w_ref.float()) < TOL
    base = ops.awq_gemm(a.to(DEV), qweight.to(DEV), scales.to(DEV), qzeros.to(DEV), 8).float().cpu()
    assert compute_max_diff(out, base) < TOL
    # the repacked codes are the Marlin layout of the same integer matrix (reference packer: marlin_weights)
    codes = torch.from_numpy(np.stack([(qweight.numpy().astype(np.uint32) >> (4 * packing._AWQ_ORDER[j])) & 0xf for j in range(8)],
                                      axis=2).reshape(k, n).astype(np.int32))
    assert torch.equal(mq.cpu(), packing.marlin_weights(codes, k, n, 4))


@pytest.mark.parametrize("wide", ["1,4,1", "1,4,2", "1,4,4", "0", None])
@pytest.mark.parametrize("m,k,n", [(65, 512, 512), (128, 1024, 256), (200, 2048, 384), (256, 8192, 7168), (300, 3584, 1024)])
def test_awq_marlin_wide_kernel(ops, tune, m, k, n, wide):
    """Round 3: marlin_wide_kernel<ZP> (AWQ weights on the wide tiles, M > 64: the hand-placed conversion plan with its two
    fix-up constants taken from the zero row) on its 128 x 256 tiles, with K splits, ragged rows, padding column groups,
    and the 70B / TP = 8 gate_up shape - against a @ w_ref, and against the row-block kernel (NMX_GEMM_WIDE=0)."""
    if wide not in (None, "0") and (k, n) == (8192, 7168) and wide != "1,4,4":
        pytest.skip("large shape covered with one forced configuration")
    seed_all(m + k)
    w = torch.randn(k, n)
    a = torch.randn(m, k, dtype=torch.float16)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(w, 128)
    mq, ms, mz = ops.awq_marlin_repack(qweight.to(DEV), qzeros.to(DEV), scales.to(DEV))
    tune(NMX_GEMM_WIDE=wide)
    out = ops.awq_marlin_gemm(a.to(DEV), mq, ms, mz, m, n, k)
    assert compute_max_diff(out.float().cpu(), a.float() @ w_ref.float()) < TOL
    d = ops.awq_marlin_gemm_deferred(a.to(DEV), mq, ms, mz, m, n, k)
    assert torch.equal(d.materialize().view(torch.int16), out.view(torch.int16))
    tune(NMX_GEMM_WIDE="0")
    base = ops.awq_marlin_gemm(a.to(DEV), mq, ms, mz, m, n, k)
    assert compute_max_diff(out.float().cpu(), base.float().cpu()) < TOL


def test_awq_layer_uses_marlin_path(ops):
    from neuralmagic_vllm_amd.layers.linear import ColumnParallelLinear
    from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
    seed_all(5)
    K, N = 1024, 512
    w_ref, qweight, qzeros, scales = packing.awq_quantize(torch.randn(K, N), 128)
    layer = ColumnParallelLinear(K, N, AWQConfig(4, 128, True))
    for name, t in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales)):
        prm = getattr(layer, name)
        prm.weight_loader(prm, t)
        prm.data = prm.data.to(DEV)
    layer.quant_method.process_weights_after_loading(layer)
    assert layer.marlin_q is not None
    x = torch.randn(40, K, dtype=torch.float16)
    y = layer(x.to(DEV))
    assert compute_max_diff(y.cpu(), x.float() @ w_ref.float()) < TOL
