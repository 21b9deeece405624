"""CPU tests of the host-side mirror (no GPU, no kernels): quantization configs, parameter shapes / sharding attributes
created by every LinearMethod, scale permutations, op argument validation that runs before any launch."""
import pytest
import torch

from oracle import packing


class Layer(torch.nn.Module):
    pass


def test_registry_and_config_parsing():
    from neuralmagic_vllm_amd.layers.quantization import QUANTIZATION_METHODS, get_quantization_config
    assert {"awq", "fp8", "gptq", "gptq_marlin", "gptq_marlin_24", "marlin", "compressed-tensors"} <= set(QUANTIZATION_METHODS)
    with pytest.raises(ValueError):
        get_quantization_config("squeezellm")
    gm = get_quantization_config("gptq_marlin").from_config({"bits": 4, "group_size": 128, "desc_act": True, "sym": True})
    assert (gm.weight_bits, gm.group_size, gm.desc_act, gm.pack_factor) == (4, 128, True, 8)
    assert get_quantization_config("gptq").from_config({"bits": 3, "group_size": 64, "desc_act": False}).weight_bits == 3
    assert get_quantization_config("awq").from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True}).group_size == 128
    fp8 = get_quantization_config("fp8").from_config({"quant_method": "fp8", "activation_scheme": "static"})
    assert fp8.is_checkpoint_fp8_serialized and fp8.activation_scheme == "static"
    assert get_quantization_config("marlin").from_config({"group_size": -1}).group_size == -1


def test_gptq_marlin_compatibility_and_override():
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig
    ok = {"quant_method": "gptq", "bits": 4, "group_size": 128, "sym": True, "desc_act": False}
    assert GPTQMarlinConfig.is_marlin_compatible(ok)
    assert not GPTQMarlinConfig.is_marlin_compatible({**ok, "bits": 3})
    assert not GPTQMarlinConfig.is_marlin_compatible({**ok, "sym": False})
    assert not GPTQMarlinConfig.is_marlin_compatible({**ok, "group_size": 16})
    assert GPTQMarlinConfig.override_quantization_method(ok, None) == "gptq_marlin"
    assert GPTQMarlinConfig.override_quantization_method(ok, "awq") is None
    with pytest.raises(ValueError):
        GPTQMarlinConfig(5, 128, False, True)
    with pytest.raises(ValueError):
        GPTQMarlinConfig(4, 128, False, False)


@pytest.mark.parametrize("bits,group,desc_act", [(4, 128, False), (8, -1, False), (4, 64, True)])
def test_gptq_marlin_create_weights_shapes(bits, group, desc_act):
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig, GPTQMarlinState
    K, N = 512, 384
    method = GPTQMarlinConfig(bits, group, desc_act, True).get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N // 3] * 3, K, N, torch.float16)
    pf = 32 // bits
    assert layer.qweight.shape == (K // pf, N) and layer.qweight.pack_factor == pf and layer.qweight.packed_dim == 0
    g = 1 if group == -1 else K // group
    assert layer.scales.shape == (g, N) and layer.g_idx.shape == (K, )
    assert layer.workspace.numel() == N // 64 * 16 and int(layer.workspace.abs().sum()) == 0
    assert layer.marlin_state == GPTQMarlinState.REPACK and layer.is_k_full
    with pytest.raises(ValueError):
        method.create_weights(Layer(), K, [100], K, 100, torch.float16)  # N % 64
    with pytest.raises(ValueError):
        method.create_weights(Layer(), 192, [N], 192, N, torch.float16)  # K % 128


def test_marlin_scale_permutation_matches_reference_fixture():
    import numpy as np
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import marlin_permute_scales
    from util import from_bits, load_golden
    for name in ("marlin_b4_g128_act0", "marlin_b8_g-1_act0"):
        g = load_golden(name)
        K, N = g["q_w"].shape
        gs = int(g["group_size"])
        out = marlin_permute_scales(from_bits(g["s"], torch.float16), K, N, K if gs == -1 else gs, int(g["bits"]))
        assert np.array_equal(out.view(torch.int16).numpy().view(np.uint16), g["marlin_s"])


def test_other_methods_create_weights_shapes():
    from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
    from neuralmagic_vllm_amd.layers.quantization.gptq import GPTQConfig
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin_24 import GPTQMarlin24Config
    from neuralmagic_vllm_amd.layers.quantization.marlin import MarlinConfig
    K, N = 256, 512
    a = Layer()
    AWQConfig(4, 128, True).get_quant_method(None).create_weights(a, K, [N], K, N, torch.float16)
    assert a.qweight.shape == (K, N // 8) and a.qzeros.shape == (K // 128, N // 8) and a.scales.shape == (K // 128, N)
    g = Layer()
    GPTQConfig(3, 64, False).get_quant_method(None).create_weights(g, K, [N], K, N, torch.float16)
    assert g.qweight.shape == (K * 3 // 32, N) and g.qzeros.shape == (K // 64, N * 3 // 32)
    m = Layer()
    MarlinConfig(128).get_quant_method(None).create_weights(m, K, [N], K, N, torch.float16, device="cpu")
    assert m.B.shape == (K // 16, N * 2) and m.s.shape == (2, N)
    s = Layer()
    GPTQMarlin24Config(8, -1).get_quant_method(None).create_weights(s, K, [N], K, N, torch.float16, device="cpu")
    assert s.B_24.shape == (K // 32, N * 4) and s.B_meta.shape == (K // 32, 2 * N) and s.s.shape == (1, N)
    with pytest.raises(ValueError):
        AWQConfig(3, 128, True)


def test_ops_validate_arguments_before_touching_the_gpu():
    from neuralmagic_vllm_amd import _custom_ops as ops
    a = torch.zeros(2, 128, dtype=torch.float16)
    q = torch.zeros(8, 128, dtype=torch.int32)
    s = torch.zeros(1, 64, dtype=torch.float16)
    e = torch.empty(0, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        ops.gptq_marlin_gemm(a, q, s, e, e, torch.zeros(16, dtype=torch.int32), 4, 2, 64, 128, True)  # CPU tensors
    with pytest.raises(RuntimeError):
        ops.gptq_marlin_24_gemm(a, q, torch.zeros(4, 128, dtype=torch.int16), s, torch.zeros(64, dtype=torch.int32), 5, 2, 64, 128)
    with pytest.raises(RuntimeError):
        ops.paged_attention_v1(torch.zeros(1, 8, 64), torch.zeros(1, 8, 64), torch.zeros(1), torch.zeros(1), 8, 1.0,
                               torch.zeros(1, 1, dtype=torch.int64), torch.zeros(1, dtype=torch.int32), 16, 16, None, "auto", 1.0)


def test_packers_round_trip_every_bit_width():
    import oracle
    torch.manual_seed(0)
    for bits in (2, 3, 4, 8):
        w = torch.randn(128, 64)
        w_ref, qw, qz, s, g_idx = packing.gptq_quantize(w, bits, 32)
        assert qw.shape == (128 * bits // 32, 64) and qz.shape == (4, 64 * bits // 32)
        assert torch.equal(oracle.gptq_dequantize(qw, qz, s, None, bits), w_ref)
        assert torch.equal(oracle.gptq_dequantize(qw, qz, s, g_idx, bits), w_ref)


def test_split_plan_of_the_llama_shapes_is_stable():
    """Host-only view of the launch heuristics through the scratch-size query (no GPU): the K splits chosen for the shapes
    the heuristics were fitted on, so that a change of plan shows up in review instead of in a benchmark."""
    import ctypes

    from neuralmagic_vllm_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.nmx_marlin_gemm_scratch_bytes.restype = ctypes.c_int64
    lib.nmx_scaled_mm_scratch_bytes.restype = ctypes.c_int64

    def splits(fn, m, n, k):
        return fn(m, n, k) // (m * n * 4)

    gm, mm = lib.nmx_marlin_gemm_scratch_bytes, lib.nmx_scaled_mm_scratch_bytes
    # (K, N): qkv, o, gate_up, down of Llama-3-8B
    # (the query is the maximum over the dense and the 2:4-sparse plan: sparse qkv at 32 < M <= 64 runs 4 splits on the wide tiles)
    assert [splits(gm, 1, 6144, 4096), splits(gm, 64, 6144, 4096)] == [2, 4]
    assert [splits(gm, 1, 4096, 4096), splits(gm, 64, 4096, 4096)] == [4, 4]
    assert [splits(gm, 1, 28672, 4096), splits(gm, 64, 28672, 4096)] == [0, 0]   # gate_up: enough column tiles, no split
    assert splits(gm, 64, 4096, 14336) == 8
    assert gm(0, 4096, 4096) == 0 and mm(0, 4096, 4096) == 0
    assert [splits(mm, 64, 6144, 4096), splits(mm, 64, 28672, 4096), splits(mm, 64, 4096, 14336)] == [2, 0, 4]
    # batch 256 (the bench default): int4 / sparse qkv, o, gate_up, down and the fp8 tile kernel's plan
    assert [splits(gm, 256, n, k) for n, k in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336))] == [2, 4, 0, 8]
    assert [splits(mm, 256, n, k) for n, k in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336))] == [2, 4, 0, 8]


def test_small_batch_host_rules():
    """Host-only rules added in round 3's second session (no GPU): the partition size paged_attention_v2 runs with, the shapes the
    norm-fused / attention-reduce GEMM forms serve by default, the counter buffer size."""
    import ctypes

    from neuralmagic_vllm_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.nmx_paged_attention_counters_numel.restype = ctypes.c_int64
    ps = lib.nmx_paged_attention_partition_size
    # Llama-3-8B heads (32 / 8): finer partitions while one round of workgroups holds them, >= 128 tokens, <= 8 partitions per sequence
    assert [ps(b, 32, 8, 1024) for b in (1, 4, 8, 16, 32, 256)] == [128, 128, 256, 512, 512, 512]
    assert [ps(1, 32, 8, c) for c in (128, 512, 2048, 4096, 8192, 32768)] == [128, 128, 256, 512, 512, 512]
    assert ps(1, 40, 2, 1024) == 128 and ps(0, 32, 8, 1024) == 512 and ps(2, 12, 12, 1024) == 128
    assert lib.nmx_paged_attention_counters_numel(4, 32, 8) == 32 and lib.nmx_paged_attention_counters_numel(3, 40, 2) == 12
    norm = lib.nmx_gptq_marlin_gemm_norm_supported  # (m, n, k, groups, bits, dtype, with_act)
    assert norm(1, 6144, 4096, 32, 4, 1, 0) == 1 and norm(1, 28672, 4096, 32, 4, 1, 1) == 1 and norm(1, 6144, 4096, 32, 4, 2, 0) == 1
    assert norm(2, 6144, 4096, 32, 4, 1, 0) == 0            # one row by default
    assert norm(1, 6144, 4096, 32, 8, 1, 0) == 0            # 4 bits only
    assert norm(1, 4096, 14336, 112, 4, 1, 0) == 0          # long K is not on the decode kernel
    assert norm(1, 1280, 8192, 64, 4, 1, 0) == 0            # hidden 8192: the norm kernel's 512 threads do not fit the 4-wave shape
    attn = lib.nmx_gptq_marlin_gemm_attn_supported  # (m, n, k, groups, bits, dtype, heads, head_size, max_parts)
    assert attn(1, 4096, 4096, 32, 4, 1, 32, 128, 8) == 0   # off by default (level with the reduce launch)
