"""GPU test of the checkpoint -> device pipeline end to end: a synthetic GPTQ (int4, g128) Llama layer written as a
Hugging Face style safetensors checkpoint (separate q/k/v, gate/up tensors) is loaded through LlamaDecoderStack, moved to
the GPU, and every projection is run through its quantization method (gptq_marlin_repack + gptq_marlin_gemm). Expected
values: a.float() @ w_ref.float() per logical matrix with w_ref from the oracle's quantizer."""
import pytest
import torch

from oracle import packing
from util import compute_max_diff, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, I, NH, NKV, G = 512, 1024, 8, 2, 128
D = H // NH


def test_gptq_checkpoint_to_marlin_kernels(tmp_path):
    from safetensors.torch import save_file

    from neuralmagic_vllm_amd.layers.quantization import get_quantization_config
    from neuralmagic_vllm_amd.model_loader import LlamaDecoderStack, safetensors_weights_iterator
    seed_all(0)
    parts = {"self_attn.q_proj": (H, NH * D), "self_attn.k_proj": (H, NKV * D), "self_attn.v_proj": (H, NKV * D),
             "self_attn.o_proj": (NH * D, H), "mlp.gate_proj": (H, I), "mlp.up_proj": (H, I), "mlp.down_proj": (I, H)}
    ck, ref = {}, {}
    for name, (k, n) in parts.items():
        w = torch.randn(k, n, dtype=torch.float16)
        w_ref, q_w, s, _, _ = packing.quantize_weights(w, 4, G, False)
        p = "model.layers.0." + name
        ck[p + ".qweight"] = packing.gptq_pack(q_w, 4, k, n).contiguous()
        ck[p + ".scales"] = s.contiguous()
        ck[p + ".g_idx"] = (torch.arange(k, dtype=torch.int32) // G)
        ck[p + ".qzeros"] = torch.full((k // G, n // 8), 0x77777777, dtype=torch.int32)
        ref[name] = w_ref.float()
    path = str(tmp_path / "model.safetensors")
    save_file(ck, path)

    cfg = get_quantization_config("gptq_marlin").from_config({"bits": 4, "group_size": G, "desc_act": False, "sym": True})
    stack = LlamaDecoderStack(H, I, NH, NKV, 1, cfg)
    assert stack.load_weights(safetensors_weights_iterator([path])) == 7 * 4
    for p in stack.parameters():
        if not p.is_meta:
            p.data = p.data.to(DEV)
    layer = stack.model.layers[0]
    x = torch.randn(24, H, dtype=torch.float16)
    qkv = layer.self_attn.qkv_proj(x.to(DEV)).cpu()
    want = torch.cat([x.float() @ ref["self_attn.q_proj"], x.float() @ ref["self_attn.k_proj"], x.float() @ ref["self_attn.v_proj"]], 1)
    assert qkv.shape == want.shape and compute_max_diff(qkv, want) < 1e-3
    gu = layer.mlp.gate_up_proj(x.to(DEV)).cpu()
    want = torch.cat([x.float() @ ref["mlp.gate_proj"], x.float() @ ref["mlp.up_proj"]], 1)
    assert compute_max_diff(gu, want) < 1e-3
    a = torch.randn(24, I, dtype=torch.float16)
    assert compute_max_diff(layer.mlp.down_proj(a.to(DEV)).cpu(), a.float() @ ref["mlp.down_proj"]) < 1e-3
    assert compute_max_diff(layer.self_attn.o_proj(x.to(DEV)).cpu(), x.float() @ ref["self_attn.o_proj"]) < 1e-3
