"""CPU tests of the checkpoint -> parameter pipeline: safetensors iteration, quantization config
discovery, stacked q/k/v and gate/up loading with TP sharding of packed dimensions, fp8 scales, KV-cache scale files.
Mirrors what vllm/model_executor/models/llama.py:433-519 and model_loader/weight_utils.py do for the hot-path layers."""
import json
import os

import pytest
import torch

from neuralmagic_vllm_amd.layers import linear as linear_mod
from neuralmagic_vllm_amd.layers.quantization import get_quantization_config
from neuralmagic_vllm_amd.model_loader import (LlamaDecoderStack, get_quant_config, kv_cache_scales_loader,
                                               safetensors_weights_iterator)
from neuralmagic_vllm_amd.model_loader import llama as llama_mod

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
H, I, NH, NKV, L, G = 512, 1024, 8, 2, 2, 128
D = H // NH


@pytest.fixture
def tp(monkeypatch):
    """Pretend to be rank r of a TP group of size n (the loaders only ask for these two numbers)."""

    def set_(rank, size):
        for mod in (linear_mod, llama_mod):
            monkeypatch.setattr(mod, "get_tensor_model_parallel_rank", lambda: rank)
            monkeypatch.setattr(mod, "get_tensor_model_parallel_world_size", lambda: size)

    return set_


def gptq_checkpoint(seed=0):
    g = torch.Generator().manual_seed(seed)
    ri = lambda *s: torch.randint(-2**31, 2**31 - 1, s, dtype=torch.int32, generator=g)  # noqa: E731
    t = {}
    for li in range(L):
        p = f"model.layers.{li}."
        for name, (k, n) in {"self_attn.q_proj": (H, NH * D), "self_attn.k_proj": (H, NKV * D), "self_attn.v_proj": (H, NKV * D),
                             "self_attn.o_proj": (NH * D, H), "mlp.gate_proj": (H, I), "mlp.up_proj": (H, I),
                             "mlp.down_proj": (I, H)}.items():
            t[p + name + ".qweight"] = ri(k // 8, n)
            t[p + name + ".qzeros"] = ri(k // G, n // 8)
            t[p + name + ".scales"] = torch.rand(k // G, n, generator=g).half()
            t[p + name + ".g_idx"] = (torch.arange(k, dtype=torch.int32) // G)
            t[p + name + ".bias"] = torch.zeros(n, dtype=torch.float16)  # GPTQ checkpoints carry an unused bias
        t[p + "input_layernorm.weight"] = torch.rand(H, generator=g).half()
        t[p + "post_attention_layernorm.weight"] = torch.rand(H, generator=g).half()
        t[p + "self_attn.rotary_emb.inv_freq"] = torch.rand(D // 2, generator=g)
    t["model.embed_tokens.weight"] = torch.rand(16, H, generator=g).half()  # not part of the hot path: skipped
    return t


def write_model_dir(tmp_path, tensors, config):
    from safetensors.torch import save_file
    names = sorted(tensors)
    half = len(names) // 2
    save_file({k: tensors[k].contiguous() for k in names[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: tensors[k].contiguous() for k in names[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    for fname, doc in config.items():
        (tmp_path / fname).write_text(json.dumps(doc))
    return sorted(str(p) for p in tmp_path.glob("*.safetensors"))


def test_gptq_checkpoint_loads_into_fused_sharded_parameters(tmp_path, tp):
    ck = gptq_checkpoint()
    files = write_model_dir(tmp_path, ck, {"quantize_config.json": {"bits": 4, "group_size": G, "desc_act": False, "sym": True}})
    cfg = get_quant_config(str(tmp_path), "gptq_marlin")
    assert (cfg.weight_bits, cfg.group_size) == (4, G)
    for size in (1, 2):
        for rank in range(size):
            tp(rank, size)
            m = LlamaDecoderStack(H, I, NH, NKV, L, cfg)
            used = m.load_weights(safetensors_weights_iterator(files))
            assert used == L * (7 * 4 + 2)  # 4 tensors per linear part + 2 norm weights; bias / rotary / embeddings skipped
            for li in range(L):
                p, lay = f"model.layers.{li}.", m.model.layers[li]
                sl = lambda x, n: x[:, rank * (n // size):(rank + 1) * (n // size)]  # noqa: E731  column shard of width n/size
                q, k, v = (ck[p + f"self_attn.{x}_proj.qweight"] for x in "qkv")
                assert torch.equal(lay.self_attn.qkv_proj.qweight.data, torch.cat([sl(q, NH * D), sl(k, NKV * D), sl(v, NKV * D)], 1))
                s = torch.cat([sl(ck[p + f"self_attn.{x}_proj.scales"], n) for x, n in (("q", NH * D), ("k", NKV * D), ("v", NKV * D))], 1)
                assert torch.equal(lay.self_attn.qkv_proj.scales.data, s)
                gu = torch.cat([sl(ck[p + "mlp.gate_proj.qweight"], I), sl(ck[p + "mlp.up_proj.qweight"], I)], 1)
                assert torch.equal(lay.mlp.gate_up_proj.qweight.data, gu)
                # row-parallel: K is split, in packed rows (8 k per int32) for qweight and in groups for the scales
                kr = (I // 8) // size
                assert torch.equal(lay.mlp.down_proj.qweight.data, ck[p + "mlp.down_proj.qweight"][rank * kr:(rank + 1) * kr])
                gr = (I // G) // size
                assert torch.equal(lay.mlp.down_proj.scales.data, ck[p + "mlp.down_proj.scales"][rank * gr:(rank + 1) * gr])
                assert torch.equal(lay.self_attn.o_proj.g_idx.data, ck[p + "self_attn.o_proj.g_idx"][rank * (H // size):(rank + 1) * (H // size)])
                assert torch.equal(lay.input_layernorm.weight.data, ck[p + "input_layernorm.weight"])


def test_kv_heads_replicated_when_tp_exceeds_kv_heads(tp):
    cfg = get_quantization_config("gptq_marlin").from_config({"bits": 4, "group_size": G, "desc_act": False, "sym": True})
    ck = gptq_checkpoint(1)
    for rank in range(4):  # 2 kv heads on 4 ranks: ranks 2 r and 2 r + 1 share kv head r (config.py:396-404)
        tp(rank, 4)
        m = LlamaDecoderStack(H, I, NH, NKV, 1, cfg)
        m.load_weights((k, v) for k, v in ck.items() if k.startswith("model.layers.0."))
        qkv = m.model.layers[0].self_attn.qkv_proj
        assert (qkv.num_heads, qkv.num_kv_heads, qkv.num_kv_head_replicas) == (2, 1, 2)
        kw = ck["model.layers.0.self_attn.k_proj.qweight"]
        head = rank // 2
        assert torch.equal(qkv.qweight.data[:, 2 * D:3 * D], kw[:, head * D:(head + 1) * D])


def test_fused_checkpoint_tensor_is_split_in_packed_and_marlin_units(tp):
    """A checkpoint that already stores gate_up fused (loaded_shard_id None): the parts are cut in units of the packed /
    Marlin-tiled output dimension (linear.py:403-421)."""
    tp(1, 2)
    cfg = get_quantization_config("marlin").from_config({"group_size": 128})
    lin = linear_mod.MergedColumnParallelLinear(H, [I, I], cfg, device="cpu")
    B = lin.B  # [K/16, N_rank * 16 / 8] int32, packed along the output dim with marlin_tile_size 16
    assert (B.packed_dim, B.output_dim, B.marlin_tile_size, B.pack_factor) == (1, 1, 16, 8)
    full = torch.arange((H // 16) * (2 * I * 16 // 8), dtype=torch.int32).reshape(H // 16, 2 * I * 16 // 8)
    lin.weight_loader(B, full)
    part = I * 16 // 8  # packed columns of one logical matrix; rank 1 takes the second half of each
    want = torch.cat([full[:, part // 2:part], full[:, part + part // 2:2 * part]], 1)
    assert torch.equal(B.data.cpu(), want)


def test_fp8_checkpoint_scales_and_kv_scale(tp):
    tp(0, 1)
    cfg = get_quantization_config("fp8").from_config({"quant_method": "fp8", "activation_scheme": "static"})
    m = LlamaDecoderStack(H, I, NH, NKV, 1, cfg, kv_cache_dtype="fp8")
    p = "model.layers.0."
    w = lambda n, k: (torch.randn(n, k) * 0.1).to(torch.float8_e4m3fn)  # noqa: E731
    ck = {}
    for i, (name, (n, k)) in enumerate({"self_attn.q_proj": (NH * D, H), "self_attn.k_proj": (NKV * D, H), "self_attn.v_proj": (NKV * D, H),
                                        "self_attn.o_proj": (H, NH * D), "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H),
                                        "mlp.down_proj": (H, I)}.items()):
        ck[p + name + ".weight"] = w(n, k)
        ck[p + name + ".weight_scale"] = torch.tensor(0.01 * (i + 1))            # AutoFP8: 0-dim
        ck[p + name + ".input_scale"] = torch.tensor([0.5 + 0.1 * i])            # compressed-tensors style: shape [1]
    ck[p + "self_attn.kv_scale"] = torch.tensor(0.023)
    m.load_weights(ck.items())
    qkv = m.model.layers[0].self_attn.qkv_proj
    assert torch.allclose(qkv.weight_scale.data, torch.tensor([0.01, 0.02, 0.03]))
    assert torch.allclose(qkv.input_scale.data, torch.tensor([0.5, 0.6, 0.7]))
    assert torch.equal(qkv.weight.data.view(torch.uint8)[:NH * D], ck[p + "self_attn.q_proj.weight"].view(torch.uint8))
    # the KV-cache method's post-load step (the linear method's one requantises on the GPU: tests/test_layers_gpu.py)
    attn = m.model.layers[0].self_attn.attn
    attn.quant_method.process_weights_after_loading(attn)
    assert attn._kv_scale == pytest.approx(0.023, rel=1e-3) and not hasattr(attn, "kv_scale")


def test_kv_cache_scales_file_of_the_reference(tp):
    path = os.path.join(GOLDEN, "kv_cache_scales_llama2_7b_fp8.json")  # tests/fp8_kv/llama2-7b-fp8-kv/kv_cache_scales.json
    doc = json.load(open(path))
    items = dict(kv_cache_scales_loader(path, 0, 1, 32, "llama"))
    assert len(items) == 32 and items[0] == pytest.approx(doc["kv_cache"]["scaling_factor"]["0"]["0"])
    # every failed check falls back to "no scales" = 1.0 everywhere (weight_utils.py:416-428)
    assert kv_cache_scales_loader(path, 0, 2, 32, "llama") == []        # TP size of the file is 1
    assert kv_cache_scales_loader(path, 0, 1, 40, "llama") == []        # layer count
    assert kv_cache_scales_loader(path, 0, 1, 32, "mistral") == []      # model type
    assert kv_cache_scales_loader(path + ".missing", 0, 1, 32, "llama") == []
    tp(0, 1)
    cfg = get_quantization_config("fp8").from_config({"quant_method": "fp8", "activation_scheme": "dynamic"})
    m = LlamaDecoderStack(H, I, NH, NKV, 32, cfg, kv_cache_dtype="fp8")
    m.load_kv_cache_scales(path)
    assert m.model.layers[31].self_attn.attn._kv_scale == pytest.approx(doc["kv_cache"]["scaling_factor"]["0"]["31"])


def test_quant_config_from_hf_config_json(tmp_path):
    (tmp_path / "config.json").write_text(json.dumps({"model_type": "llama", "quantization_config": {
        "quant_method": "awq", "w_bit": 4, "q_group_size": 128, "zero_point": True}}))
    cfg = get_quant_config(str(tmp_path), "awq")
    assert (cfg.weight_bits, cfg.group_size, cfg.zero_point) == (4, 128, True)
    (tmp_path / "config.json").write_text(json.dumps({"model_type": "llama"}))
    with pytest.raises(ValueError):
        get_quant_config(str(tmp_path), "awq")  # no quant_config.json either
