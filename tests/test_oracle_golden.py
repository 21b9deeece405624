"""CPU tests: pins the oracle (oracle/) against golden vectors produced by the REFERENCE itself
(tests/golden/gen_golden.py: reference CPU backend csrc/cpu + reference Python quant utilities).

The oracle follows the reference's CUDA algorithms (e.g. probabilities rounded to scalar_t before P.V, the
1/(sum+1e-6) normaliser) while the reference CPU backend keeps fp32 probabilities and divides by the plain sum, so
attention agrees to rounding, not bit-for-bit; integer/byte work (cache ops, packing) must agree exactly.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import packing
from util import DTYPES, from_bits, load_golden


@pytest.mark.parametrize("name", ["attn_f32_gqa", "attn_bf16_gqa", "attn_bf16_alibi_mha", "attn_bf16_opt125m"])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_attention_oracle_vs_reference_cpu(name, version):
    g = load_golden(name)
    dt = DTYPES[str(g["dtype"])]
    q = from_bits(g["q"], dt)
    kc = from_bits(g["k_cache"], dt)
    vc = from_bits(g["v_cache"], dt)
    bt = torch.from_numpy(g["block_tables"])
    sl = torch.from_numpy(g["seq_lens"])
    al = torch.from_numpy(g["alibi_slopes"]) if "alibi_slopes" in g.files else None
    scale, kvh, max_len = float(g["scale"]), int(g["num_kv_heads"]), int(g["max_seq_len"])
    out = torch.empty_like(q)
    if version == "v1":
        oracle.paged_attention_v1(out, q, kc, vc, kvh, scale, bt, sl, 16, max_len, al, "auto", 1.0)
    else:
        S, H, D = q.shape
        P = (max_len + 511) // 512
        tmp = torch.empty(S, H, P, D, dtype=dt)
        es = torch.empty(S, H, P, dtype=torch.float32)
        ml = torch.empty(S, H, P, dtype=torch.float32)
        oracle.paged_attention_v2(out, es, ml, tmp, q, kc, vc, kvh, scale, bt, sl, 16, max_len, al, "auto", 1.0)
    ref = from_bits(g["out_" + version], dt)
    # reference bar (tests/kernels/test_attention.py:281-284): atol 1e-3, rtol 1e-5
    atol = 2e-5 if dt == torch.float32 else 1e-3
    torch.testing.assert_close(out.float(), ref.float(), atol=atol, rtol=1e-5 if dt != torch.float32 else 1e-4)


@pytest.mark.parametrize("tag", ["f32", "bf16", "bf16_opt125m"])
def test_reshape_and_cache_oracle_bit_exact(tag):
    g = load_golden("reshape_and_cache_" + tag)
    dt = DTYPES[tag.split("_")[0]]
    qkv = from_bits(g["qkv"], dt)
    key, value = qkv[:, 1], qkv[:, 2]
    kc = from_bits(g["k_cache_in"], dt)
    vc = from_bits(g["v_cache_in"], dt)
    oracle.reshape_and_cache(key, value, kc, vc, torch.from_numpy(g["slot_mapping"]), "auto", 1.0)
    assert torch.equal(kc.view(torch.int16 if dt != torch.float32 else torch.int32),
                       from_bits(g["k_cache_out"], dt).view(torch.int16 if dt != torch.float32 else torch.int32))
    assert torch.equal(vc.view(torch.int16 if dt != torch.float32 else torch.int32),
                       from_bits(g["v_cache_out"], dt).view(torch.int16 if dt != torch.float32 else torch.int32))


def test_copy_blocks_oracle_bit_exact():
    g = load_golden("copy_blocks_f32")
    kcs = [torch.from_numpy(a.copy()) for a in g["k_in"]]
    vcs = [torch.from_numpy(a.copy()) for a in g["v_in"]]
    oracle.copy_blocks(kcs, vcs, torch.from_numpy(g["mapping"]))
    for l in range(len(kcs)):
        assert np.array_equal(kcs[l].numpy(), g["k_out"][l])
        assert np.array_equal(vcs[l].numpy(), g["v_out"][l])


MARLIN_CASES = ["marlin_b4_g128_act0", "marlin_b4_g-1_act0", "marlin_b4_g64_act1", "marlin_b8_g128_act0",
                "marlin_b8_g-1_act0", "marlin_b8_g32_act1"]


@pytest.mark.parametrize("name", MARLIN_CASES)
def test_packers_match_reference_utils(name):
    """numpy restatement (oracle/packing.py) == reference Python utilities, bit for bit."""
    g = load_golden(name)
    bits, gs = int(g["bits"]), int(g["group_size"])
    w = from_bits(g["w"], torch.float16)
    K, N = w.shape
    act = g["g_idx"].size > 0
    q_w = torch.from_numpy(g["q_w"].astype(np.int32))
    # quantize_weights (deterministic part: before the random act-order permutation)
    w_ref, q, s, _, _ = packing.quantize_weights(w, bits, K if gs == -1 else gs, False)
    if not act:
        assert torch.equal(q, q_w)
        assert torch.equal(w_ref.view(torch.int16), from_bits(g["w_ref"], torch.float16).view(torch.int16))
    else:
        rp = torch.from_numpy(g["rand_perm"])
        assert torch.equal(q[rp], q_w)
    assert torch.equal(s.view(torch.int16), from_bits(g["s"], torch.float16).view(torch.int16))
    assert np.array_equal(packing.gptq_pack(q_w, bits, K, N).numpy(), g["q_gptq"])
    q_sorted = q_w
    if act:
        q_sorted, g_sorted, sort_idx = packing.sort_weights(q_w, torch.from_numpy(g["g_idx"]))
        assert np.array_equal(g_sorted.numpy(), g["g_idx_sorted"])
        # argsort of equal keys is not unique: compare the sorted weights through the golden sort order instead
        q_sorted = q_w[torch.from_numpy(g["sort_idx"]).long()]
    assert np.array_equal(packing.marlin_weights(q_sorted, K, N, bits).numpy(), g["marlin_q"])
    ms = packing.marlin_permute_scales(from_bits(g["s"], torch.float16), K, N, K if gs == -1 else gs)
    assert np.array_equal(ms.view(torch.int16).numpy().view(np.uint16), g["marlin_s"])


@pytest.mark.parametrize("name", MARLIN_CASES)
def test_marlin_unpack_and_gemm_oracle(name):
    """C++ oracle: Marlin layout -> codes (exact) and dequant-GEMM vs the reference's fake-quant matmul."""
    g = load_golden(name)
    bits, gs = int(g["bits"]), int(g["group_size"])
    K, N = g["q_w"].shape
    act = g["g_idx"].size > 0
    marlin_q = torch.from_numpy(g["marlin_q"])
    q_sorted = g["q_w"]
    if act:
        q_sorted = g["q_w"][g["sort_idx"].astype(np.int64)]
    assert np.array_equal(oracle.marlin_unpack(marlin_q, K, N, bits).numpy(), q_sorted)

    marlin_s = from_bits(g["marlin_s"], torch.float16)
    a = from_bits(g["a"], torch.float16)
    g_idx = torch.from_numpy(g["g_idx_sorted"]) if act else torch.empty(0, dtype=torch.int32)
    sort_idx = torch.from_numpy(g["sort_idx"]) if act else torch.empty(0, dtype=torch.int32)
    if act:
        # the reference test multiplies the ORIGINAL activations with w_ref in the random-permuted row order
        # (test_marlin_gemm.py:153-172): a @ w_ref where row i of w_ref is original row rand_perm[i]. The kernel
        # receives weights sorted by group (rows sort_idx of that), and gathers A columns by sort_idx.
        pass
    c = oracle.gptq_marlin_gemm(a, marlin_q, marlin_s, g_idx, sort_idx, None, bits, a.shape[0], N, K, True)
    c_ref = torch.from_numpy(g["c_ref"])
    err = float((c.float() - c_ref).abs().mean() / c_ref.abs().mean())
    assert err < 1e-3, err  # only fp16 output rounding separates the two


MARLIN24_CASES = ["marlin24_b4_g-1", "marlin24_b4_g128", "marlin24_b8_g128"]


@pytest.mark.parametrize("name", MARLIN24_CASES)
def test_marlin_24_restatement_matches_reference_utils(name):
    """2:4 pipeline (mask_creator -> quantize -> CUTLASS compress + meta reorder -> Marlin-24 permutation) restated
    in numpy is bit-identical to the reference's utilities on the same weights, and the inverse (decode) recovers
    the reference's w_24_ref exactly; the oracle GEMM then matches the reference's a @ w_24_ref."""
    g = load_golden(name)
    bits, gs = int(g["bits"]), int(g["group_size"])
    w = from_bits(g["w"], torch.float16)
    K, N = w.shape
    mask = packing.mask_creator(w.t()).t().bool()
    assert np.array_equal(mask.numpy(), g["mask"])
    w_ref, mq, meta, ms = packing.marlin_24_quantize(w, bits, gs)
    assert np.array_equal(w_ref.view(torch.int16).numpy().view(np.uint16), g["w_24_ref"])
    assert np.array_equal(mq.numpy(), g["marlin_24_q"])
    assert np.array_equal(meta.numpy(), g["meta"])
    assert np.array_equal(ms.view(torch.int16).numpy().view(np.uint16), g["marlin_24_s"])
    wd = packing.marlin_24_decode(torch.from_numpy(g["marlin_24_q"]), torch.from_numpy(g["meta"]),
                                  from_bits(g["marlin_24_s"], torch.float16), bits, K, N, gs)
    assert np.array_equal(wd.view(torch.int16).numpy().view(np.uint16), g["w_24_ref"])
    a = from_bits(g["a"], torch.float16)
    c = oracle.gptq_marlin_24_gemm(a, torch.from_numpy(g["marlin_24_q"]), torch.from_numpy(g["meta"]),
                                   from_bits(g["marlin_24_s"], torch.float16), None, bits, a.shape[0], N, K)
    c_ref = torch.from_numpy(g["c_ref"])
    err = float((c.float() - c_ref).abs().mean() / c_ref.abs().mean())
    assert err < 1e-3, err


def test_pack_fp8_matches_reference():
    g = load_golden("pack_fp8")
    w8 = torch.from_numpy(g["w8"]).view(torch.float8_e4m3fn)
    assert np.array_equal(packing.pack_fp8_to_int32(w8).numpy(), g["packed"])


def test_fp8_e4m3_conversion_matches_torch():
    """oracle number formats vs torch's own float8 casts (all 256 codes + random round trips)."""
    codes = torch.arange(256, dtype=torch.int32).to(torch.uint8)
    f = codes.view(torch.float8_e4m3fn).float()
    x = torch.randn(4096) * 50
    x[:16] = torch.tensor([0.0, -0.0, 448.0, -448.0, 464.0, 1000.0, -1e6, 2**-9, 2**-10, 1.5 * 2**-9, 0.0146, 0.0156,
                           0.0157, 239.9, 240.1, 17.0])
    q, s = oracle.scaled_fp8_quant(x, torch.tensor([1.0]))
    expect = x.clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q.view(torch.uint8), expect.view(torch.uint8))
    # decode path: convert_fp8(fp8 -> f32) on every finite code
    out = torch.empty(256, dtype=torch.float32)
    oracle.convert_fp8(out, codes, 1.0, "fp8")
    finite = ~torch.isnan(f)
    assert torch.equal(out[finite], f[finite])
    # e5m2
    f5 = codes.view(torch.float8_e5m2).float()
    out5 = torch.empty(256, dtype=torch.float32)
    oracle.convert_fp8(out5, codes, 1.0, "fp8_e5m2")
    fin5 = torch.isfinite(f5)
    assert torch.equal(out5[fin5], f5[fin5])


def test_half_conversion_matches_torch():
    x = torch.cat([torch.randn(10000) * 100, torch.randn(10000) * 1e-6, torch.tensor([65504.0, 65519.0, 65520.0, 1e-8])])
    a = torch.empty(x.numel(), dtype=torch.float16)
    w = torch.eye(1, dtype=torch.float32)
    # matmul with a 1x1 identity rounds each element through the oracle's float->half
    y = oracle.matmul(x.reshape(-1, 1), w, out_dtype=torch.float16).reshape(-1)
    assert torch.equal(y.view(torch.int16), x.to(torch.float16).view(torch.int16))
    yb = oracle.matmul(x.reshape(-1, 1), w, out_dtype=torch.bfloat16).reshape(-1)
    assert torch.equal(yb.view(torch.int16), x.to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("window,use_alibi", [(0, False), (16, False), (0, True)])
def test_prefix_prefill_oracle_matches_reference_test_expectation(window, use_alibi):
    """The reference pins context_attention_fwd by comparing with plain causal attention over context + new tokens
    (tests/kernels/test_prefix_prefill.py:160-215, xformers BlockDiagonalCausalFromBottomRightMask [+ local window]).
    The oracle (C++ restatement of the Triton kernel's arithmetic) must agree with a torch fp32 statement of that."""
    import random
    from util import ref_prefix_prefill, seed_all
    seed_all(0)
    batch, H, Hkv, D, BS = 4, 8, 2, 64, 16
    q_lens = [random.randint(1, 40) for _ in range(batch)]
    ctx = [0] + [random.randint(1, 50) for _ in range(batch - 1)]
    T = sum(q_lens)
    q, k, v = (torch.empty(T, h, D).uniform_(-1, 1) for h in (H, Hkv, Hkv))
    kc = torch.empty(64, Hkv, D // 8, BS, 8).uniform_(-1, 1)
    vc = torch.empty(64, Hkv, D, BS).uniform_(-1, 1)
    b_loc = torch.randperm(64)[:batch * 5].reshape(batch, 5).to(torch.int32)
    start = torch.cumsum(torch.tensor([0] + q_lens[:-1]), 0).to(torch.int32)
    sl = torch.tensor([a + b for a, b in zip(q_lens, ctx)], dtype=torch.int32)
    cl = torch.tensor(ctx, dtype=torch.int32)
    al = torch.tensor([2.0**-(i + 1) for i in range(H)]) if use_alibi else None
    o = torch.zeros_like(q)
    oracle.context_attention_fwd(q, k, v, o, kc, vc, b_loc, start, sl, cl, max(q_lens), al, window)
    ref = ref_prefix_prefill(q, k, v, kc, vc, b_loc, start, sl, cl, al, window)
    assert float((o - ref).abs().max()) < 1e-5
