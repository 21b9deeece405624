"""GPU tests of the host-side QuantizeMethodBase mirrors (create_weights -> load checkpoint tensors -> apply) and of the
PagedAttention caller shim. Expected values: a.float() @ w_ref.float() with w_ref from the oracle packers."""
import pytest
import torch

import oracle
from oracle import packing
from util import compute_max_diff, create_kv_caches_with_random, ref_single_query_cached_kv_attention, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Layer(torch.nn.Module):
    pass


def to_dev(layer):
    """What the reference's model loader does: materialise every real parameter on the GPU (qzeros of the Marlin path
    stays on the meta device, gptq_marlin.py:333-343)."""
    for name, p in list(layer.named_parameters()):
        if not p.is_meta:
            p.data = p.data.to(DEV)


def load(layer, **tensors):
    for name, t in tensors.items():
        p = getattr(layer, name)
        assert p.shape == t.shape, (name, p.shape, t.shape)
        p.data.copy_(t)


@pytest.mark.parametrize("bits,group,desc_act", [(4, 128, False), (4, -1, False), (8, 128, False), (4, 64, True)])
@pytest.mark.parametrize("m", [1, 48])
def test_gptq_marlin_method(bits, group, desc_act, m):
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin import GPTQMarlinConfig, GPTQMarlinState
    seed_all(0)
    K, N = 512, 384
    cfg = GPTQMarlinConfig.from_config({"bits": bits, "group_size": group, "desc_act": desc_act, "sym": True})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [256, 128], K, N, torch.float16)
    assert layer.qweight.packed_dim == 0 and layer.qweight.pack_factor == 32 // bits and layer.scales.output_dim == 1
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, q_w, s, g_idx, _ = packing.quantize_weights(w, bits, K if group == -1 else group, desc_act)
    if not desc_act:
        g_idx = torch.arange(K, dtype=torch.int32) // (K if group == -1 else group)
    load(layer, qweight=packing.gptq_pack(q_w, bits, K, N), scales=s, g_idx=g_idx)
    to_dev(layer)
    x = torch.randn(3, m, K, dtype=torch.float16)
    bias = torch.randn(N, dtype=torch.float16)
    assert layer.marlin_state == GPTQMarlinState.REPACK
    out = method.apply(layer, x.to(DEV), bias.to(DEV))
    assert layer.marlin_state == GPTQMarlinState.READY and out.shape == (3, m, N)
    ref = x.float() @ w_ref.float() + bias.float()
    assert compute_max_diff(out.cpu(), ref) < 1e-3
    out2 = method.apply(layer, x.to(DEV), bias.to(DEV))  # second call: no repack
    assert torch.equal(out, out2)
    # merged gate | up projection + SiluAndMul as one op (extension): the same values as the two calls
    from neuralmagic_vllm_amd import _custom_ops as ops
    gate_up = method.apply(layer, x.to(DEV))
    two = torch.empty(3, m, N // 2, dtype=torch.float16, device=DEV)
    ops.silu_and_mul(two, gate_up)
    one = method.apply_silu_and_mul(layer, x.to(DEV))
    assert one.shape == two.shape and torch.equal(one, two)


def test_marlin_checkpoint_method():
    from neuralmagic_vllm_amd.layers.quantization.marlin import MarlinConfig
    seed_all(1)
    K, N = 256, 512
    cfg = MarlinConfig.from_config({"group_size": 128})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16, device=DEV)
    assert layer.B.marlin_tile_size == 16 and layer.B.shape == (K // 16, N * 2)
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, mq, ms, _, _, _ = packing.marlin_quantize(w, 4, 128, False)
    load(layer, B=mq.to(DEV), s=ms.to(DEV))
    x = torch.randn(5, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    assert compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 1e-3
    with pytest.raises(ValueError):
        MarlinConfig(64)


@pytest.mark.parametrize("bits,group", [(4, 128), (8, -1)])
def test_gptq_marlin_24_method(bits, group):
    from neuralmagic_vllm_amd.layers.quantization import get_quantization_config
    from neuralmagic_vllm_amd.layers.quantization.gptq_marlin_24 import GPTQMarlin24Config
    assert get_quantization_config("gptq_marlin_24") is GPTQMarlin24Config
    assert GPTQMarlin24Config.override_quantization_method({"checkpoint_format": "marlin_24"}, None) == "gptq_marlin_24"
    seed_all(2)
    K, N = 256, 512
    cfg = GPTQMarlin24Config.from_config({"bits": bits, "group_size": group})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16, device=DEV)
    assert layer.B_24.shape == (K // 32, N * 16 // (32 // bits)) and layer.B_meta.shape == (K // 32, 2 * N)
    assert layer.workspace.numel() == N // 128 * 64
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, mq, meta, ms = packing.marlin_24_quantize(w, bits, group)
    load(layer, B_24=mq.to(DEV), B_meta=meta.to(DEV), s=ms.to(DEV))
    x = torch.randn(3, 5, K, dtype=torch.float16)
    bias = torch.randn(N, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV), bias.to(DEV))
    ref = x.float() @ w_ref.float() + bias.float()
    assert out.shape == (3, 5, N) and compute_max_diff(out.cpu(), ref) < 1e-3
    with pytest.raises(ValueError):
        GPTQMarlin24Config(4, 64)
    with pytest.raises(ValueError):
        method.create_weights(Layer(), K, [N], K, N, torch.bfloat16, device=DEV)


@pytest.mark.parametrize("m", [4, 300])
def test_awq_method(m):
    from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
    seed_all(2)
    K, N, G = 512, 256, 128
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": G, "zero_point": True})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(torch.randn(K, N), G)
    load(layer, qweight=qweight, qzeros=qzeros, scales=scales)
    layer.to(DEV)
    x = torch.randn(m, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))  # m >= 256 takes the dequantize + dense GEMM branch (awq.py:166-170)
    assert compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 2e-3


@pytest.mark.parametrize("m", [4, 300])
def test_awq_method_repacked_releases_checkpoint_layout(m):
    """process_weights_after_loading repacks onto the Marlin kernel and drops qweight / qzeros / scales (both layouts
    resident doubled the AWQ weight memory); apply() keeps working from the repacked tensors alone."""
    from neuralmagic_vllm_amd.layers.quantization.awq import AWQConfig
    seed_all(2)
    K, N, G = 512, 256, 128
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": G, "zero_point": True})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16)
    w_ref, qweight, qzeros, scales = packing.awq_quantize(torch.randn(K, N), G)
    load(layer, qweight=qweight, qzeros=qzeros, scales=scales)
    layer.to(DEV)
    method.process_weights_after_loading(layer)
    assert layer.marlin_q is not None and layer.qweight.numel() == 0 and layer.qzeros.numel() == 0 and layer.scales.numel() == 0
    x = torch.randn(m, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    assert out.shape == (m, N) and compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 2e-3
    with pytest.raises(RuntimeError, match="NMX_AWQ_KEEP_CHECKPOINT_LAYOUT"):
        method.apply(layer, x.to(DEV).bfloat16())


@pytest.mark.parametrize("desc_act", [False, True])
def test_gptq_method(desc_act):
    from neuralmagic_vllm_amd.layers.quantization.gptq import ExllamaState, GPTQConfig
    seed_all(3)
    K, N, G = 512, 256, 128
    cfg = GPTQConfig.from_config({"bits": 4, "group_size": G, "desc_act": desc_act})
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16)
    w_ref, qweight, qzeros, scales, g_idx = packing.gptq_quantize(torch.randn(K, N), 4, G)
    x = torch.randn(7, K, dtype=torch.float16)
    if desc_act:  # checkpoint rows in activation order
        perm = torch.randperm(K)
        codes = torch.stack([(qweight >> (4 * i)) & 0xf for i in range(8)], dim=1).reshape(K, N)[perm]
        qweight = packing.gptq_pack(codes, 4, K, N)
        g_idx = g_idx[perm].contiguous()
        x = x[:, perm].contiguous()
        w_ref = w_ref[perm]
    load(layer, qweight=qweight, qzeros=qzeros, scales=scales, g_idx=g_idx)
    layer.to(DEV)
    out = method.apply(layer, x.to(DEV))
    assert layer.exllama_state == ExllamaState.READY
    assert compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 1e-3


@pytest.mark.parametrize("bits", [2, 3, 8])
def test_gptq_method_other_bits(bits):
    from neuralmagic_vllm_amd.layers.quantization.gptq import ExllamaState, GPTQConfig
    seed_all(7)
    K, N, G = 256, 128, 64
    method = GPTQConfig.from_config({"bits": bits, "group_size": G, "desc_act": False}).get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, [N], K, N, torch.float16)
    assert layer.qweight.shape == (K * bits // 32, N) and layer.qzeros.shape == (K // G, N * bits // 32)
    w_ref, qweight, qzeros, scales, g_idx = packing.gptq_quantize(torch.randn(K, N), bits, G)
    load(layer, qweight=qweight, qzeros=qzeros, scales=scales, g_idx=g_idx)
    layer.to(DEV)
    x = torch.randn(5, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    assert layer.exllama_state == ExllamaState.READY
    assert compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 1e-3


@pytest.mark.parametrize("serialized,scheme", [(False, "dynamic"), (True, "static"), (True, "dynamic")])
def test_fp8_method(serialized, scheme):
    from neuralmagic_vllm_amd.layers.quantization.fp8 import Fp8Config
    seed_all(4)
    K, widths = 512, [256, 128]
    N = sum(widths)
    cfg = Fp8Config(is_checkpoint_fp8_serialized=serialized, activation_scheme=scheme)
    method = cfg.get_quant_method(None)
    layer = Layer()
    method.create_weights(layer, K, widths, K, N, torch.bfloat16)
    w = torch.randn(N, K) * 0.05
    x = torch.randn(9, K, dtype=torch.bfloat16)
    if serialized:
        scales = torch.tensor([w[:256].abs().max() / 448, w[256:].abs().max() / 448])
        wq = torch.cat([(w[:256] / scales[0]).clamp(-448, 448).to(torch.float8_e4m3fn),
                        (w[256:] / scales[1]).clamp(-448, 448).to(torch.float8_e4m3fn)])
        load(layer, weight=wq, weight_scale=scales)
        if scheme == "static":
            load(layer, input_scale=torch.tensor([0.02, 0.02]))
    else:
        load(layer, weight=w.to(torch.bfloat16))
    layer.to(DEV)
    method.process_weights_after_loading(layer)
    assert layer.weight.dtype == torch.float8_e4m3fn and layer.weight.stride(0) == 1  # column-major [K, N]
    out = method.apply(layer, x.to(DEV))
    ref = x.float() @ w.t()
    assert out.shape == (9, N) and compute_max_diff(out.cpu(), ref) < 0.06  # fp8 W and A: ~2^-4 relative per operand


def test_int8_w8a8_method():
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors_w8a8 import CompressedTensorsW8A8
    seed_all(5)
    K, widths = 256, [128, 128]
    N = sum(widths)
    for static in (False, True):
        scheme = CompressedTensorsW8A8("tensor", static)
        layer = Layer()
        scheme.create_weights(layer, widths, K, torch.float16)
        w = torch.randn(N, K) * 0.05
        ws = torch.tensor([w[:128].abs().max() / 127, w[128:].abs().max() / 127])
        wq = torch.cat([torch.round(w[:128] / ws[0]), torch.round(w[128:] / ws[1])]).to(torch.int8)
        load(layer, weight=wq, weight_scale=ws)
        if static:
            load(layer, input_scale=torch.tensor([0.03]))
        layer.to(DEV)
        scheme.process_weights_after_loading(layer)
        x = torch.randn(6, K, dtype=torch.float16)
        out = scheme.apply_weights(layer, x.to(DEV))
        assert compute_max_diff(out.cpu(), x.float() @ w.t()) < 0.03


def _ct_config(fmt, weights, acts=None):
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors import CompressedTensorsConfig
    return CompressedTensorsConfig.from_config({
        "format": fmt, "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": weights, "input_activations": acts}}})


class Linear(torch.nn.Module):  # class name is the compressed-tensors target
    pass


@pytest.mark.parametrize("bits,group", [(4, 128), (8, None)])
def test_compressed_tensors_wna16(bits, group):
    """pack-quantized int4/int8 -> CompressedTensorsWNA16 -> gptq_marlin_repack + gptq_marlin_gemm."""
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors import CompressedTensorsWNA16
    seed_all(4)
    K, N = 256, 256
    cfg = _ct_config("pack-quantized", {"num_bits": bits, "type": "int", "symmetric": True,
                                        "strategy": "group" if group else "channel", "group_size": group})
    method = cfg.get_quant_method(None)
    layer = Linear()
    method.create_weights(layer, K, [N], K, N, torch.float16, device=DEV)
    assert isinstance(layer.scheme, CompressedTensorsWNA16)
    w = torch.randn(K, N, dtype=torch.float16)
    gs = group if group else K
    w_ref, q_w, s, _, _ = packing.quantize_weights(w, bits, gs, False)
    # checkpoint layout: weight_packed [N, K/pack] (element k of a row in bits (k % pack) * bits), scales [N, K/g]
    packed = packing.gptq_pack(q_w, bits, K, N).t().contiguous()
    load(layer, weight_packed=packed.to(DEV), weight_scale=s.t().contiguous().to(DEV))
    x = torch.randn(2, 7, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    assert out.shape == (2, 7, N) and compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 1e-3
    assert torch.equal(out, method.apply(layer, x.to(DEV)))  # second call: state READY, no repack
    with pytest.raises(ValueError):
        method.apply(layer, x.to(DEV), torch.zeros(N, device=DEV))


def test_compressed_tensors_w4a16_sparse24():
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors import CompressedTensorsW4A16Sparse24
    seed_all(5)
    K, N = 256, 256
    cfg = _ct_config("marlin-24", {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "group", "group_size": 128})
    method = cfg.get_quant_method(None)
    layer = Linear()
    method.create_weights(layer, K, [N], K, N, torch.float16, device=DEV)
    assert isinstance(layer.scheme, CompressedTensorsW4A16Sparse24)
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, mq, meta, ms = packing.marlin_24_quantize(w, 4, 128)
    load(layer, weight_packed=mq.to(DEV), meta=meta.to(DEV), scale_packed=ms.to(DEV))
    x = torch.randn(9, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    assert compute_max_diff(out.cpu(), x.float() @ w_ref.float()) < 1e-3


def test_compressed_tensors_w8a8_via_config():
    from neuralmagic_vllm_amd.layers.quantization.compressed_tensors_w8a8 import CompressedTensorsW8A8
    seed_all(6)
    K, N = 128, 256
    cfg = _ct_config("int-quantized", {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "channel"},
                     {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "token", "dynamic": True})
    method = cfg.get_quant_method(None)
    layer = Linear()
    method.create_weights(layer, K, [N], K, N, torch.float16)
    assert isinstance(layer.scheme, CompressedTensorsW8A8) and not layer.scheme.is_static_input_scheme
    w = torch.randn(N, K)
    ws = w.abs().amax(dim=1, keepdim=True) / 127
    wq = torch.clamp(torch.round(w / ws), -128, 127).to(torch.int8)
    load(layer, weight=wq, weight_scale=ws)
    to_dev(layer)
    method.process_weights_after_loading(layer)
    x = torch.randn(5, K, dtype=torch.float16)
    out = method.apply(layer, x.to(DEV))
    xs = x.float().abs().amax(dim=1, keepdim=True) / 127
    xq = torch.clamp(torch.round(x.float() / xs), -128, 127)
    ref = (xq @ wq.float().t()) * xs * ws.t()
    assert compute_max_diff(out.cpu(), ref) < 2e-3
    with pytest.raises(NotImplementedError):
        _ct_config("pack-quantized", {"num_bits": 3, "type": "int", "symmetric": True, "strategy": "channel"}
                   ).get_scheme(Linear())


def test_paged_attention_shim():
    from neuralmagic_vllm_amd.attention.ops.paged_attn import PagedAttention
    seed_all(6)
    nq, nkv, D, BS, NB = 32, 8, 128, 16, 128
    kv = torch.zeros(PagedAttention.get_kv_cache_shape(NB, BS, nkv, D), dtype=torch.half, device=DEV)
    kc, vc = PagedAttention.split_kv_cache(kv, nkv, D)
    assert kc.shape == (NB, nkv, D // 8, BS, 8) and vc.shape == (NB, nkv, D, BS)
    lens = [600, 33]
    bt = torch.randperm(NB)[:2 * 38].reshape(2, 38).to(torch.int32)
    scale = D**-0.5
    # fill the cache token by token through write_to_paged_cache, keep a host copy for the expected value
    kcs, vcs = create_kv_caches_with_random(NB, BS, 1, nkv, D, "auto", torch.half)
    for s, L in enumerate(lens):
        pos = torch.arange(L)
        slots = bt[s, pos // BS].long() * BS + pos % BS
        k = kcs[0][bt[s, pos // BS].long(), :, :, pos % BS, :].reshape(L, nkv, D)
        v = vcs[0][bt[s, pos // BS].long(), :, :, pos % BS]
        PagedAttention.write_to_paged_cache(k.to(DEV), v.to(DEV), kc, vc, slots.to(DEV), "auto", 1.0)
    q = torch.empty(2, nq, D, dtype=torch.half).uniform_(-scale, scale)
    out = PagedAttention.forward_decode(q.to(DEV), kc, vc, bt.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV),
                                        max(lens), "auto", nkv, scale, None, 1.0)  # 2 seqs x 32 heads <= 512 -> v2
    ref = ref_single_query_cached_kv_attention(q, nq // nkv, kc.cpu(), vc.cpu(), bt, torch.tensor(lens), scale, None)
    torch.testing.assert_close(out.cpu().float(), ref, atol=1e-3, rtol=1e-5)
    # copy-on-write of one block in every "layer"
    kv2 = kv.clone()
    PagedAttention.copy_blocks([kv, kv2], torch.tensor([[int(bt[0, 0]), 127]], device=DEV))
    assert torch.equal(kv[:, 127], kv[:, int(bt[0, 0])]) and torch.equal(kv2[:, 127], kv2[:, int(bt[0, 0])])


def test_torch_ops_registration():
    """`import neuralmagic_vllm_amd.torch_bindings` stands in for `import vllm._C`: the reference's op names resolve and
    run the HIP kernels (vllm/_custom_ops.py calls torch.ops._C.* / _C_cache_ops.*)."""
    import neuralmagic_vllm_amd.torch_bindings  # noqa: F401
    seed_all(7)
    op, _ = torch._C._jit_get_operation("_C::gptq_marlin_gemm")  # is_custom_op_supported() of the reference
    assert op is not None
    K, N, M = 256, 128, 3
    w = torch.randn(K, N, dtype=torch.float16)
    w_ref, mq, ms, _, _, _ = packing.marlin_quantize(w, 4, 128, False)
    a = torch.randn(M, K, dtype=torch.float16)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    out = torch.ops._C.gptq_marlin_gemm(a.to(DEV), mq.to(DEV), ms.to(DEV), e, e, ws, 4, M, N, K, True)
    assert compute_max_diff(out.cpu(), a.float() @ w_ref.float()) < 1e-3
    x = torch.randn(4, 256, dtype=torch.float16, device=DEV)
    o = torch.empty(4, 128, dtype=torch.float16, device=DEV)
    torch.ops._C.silu_and_mul(o, x)
    torch.testing.assert_close(o.float(), torch.nn.functional.silu(x[:, :128].float()) * x[:, 128:].float(), atol=2e-3, rtol=2e-3)
    assert torch.ops._C_cuda_utils.get_max_shared_memory_per_block_device_attribute(0) >= 64 * 1024
    assert torch.ops._C.cutlass_scaled_mm_supports_fp8(95)
