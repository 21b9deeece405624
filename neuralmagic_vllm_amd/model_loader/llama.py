"""Llama decoder-stack weights on the quantized hot path: the module tree whose parameter names match a Hugging Face
Llama checkpoint (model.layers.N.self_attn.{q,k,v,o}_proj.*, mlp.{gate,up,down}_proj.*, *_layernorm.weight) and the
checkpoint -> parameter mapping of vllm/model_executor/models/llama.py:433-490 (stacked q/k/v and gate/up parts, GPTQ
bias skipping, fp8 `kv_scale` remapping) plus `load_kv_cache_scales` (:495-519).

Only the hot-path part of the model lives here (the four quantized linears, the attention's KV-cache scale and the two
norm weights of every layer); embeddings, lm_head, sampler and the forward pass stay with the caller (the reference's
engine, or bench.py's synthetic driver)."""
from typing import Iterable, Optional, Tuple

import torch
from torch import nn

from neuralmagic_vllm_amd.distributed import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
from neuralmagic_vllm_amd.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from neuralmagic_vllm_amd.layers.quantization.base_config import QuantizationConfig
from neuralmagic_vllm_amd.layers.quantization.fp8 import Fp8Config, Fp8KVCacheMethod
from neuralmagic_vllm_amd.model_loader.weight_utils import default_weight_loader, kv_cache_scales_loader


class AttentionState(nn.Module):
    """What vllm/attention/layer.py keeps per layer for the paged-attention ops: the KV-cache dtype and its scaling
    factor (`_kv_scale`, default 1.0; an fp8 checkpoint may carry `kv_scale`, see Fp8KVCacheMethod)."""

    def __init__(self, kv_cache_dtype: str = "auto", quant_config: Optional[QuantizationConfig] = None):
        super().__init__()
        self.kv_cache_dtype = kv_cache_dtype
        self._kv_scale = 1.0
        self.quant_method = Fp8KVCacheMethod(quant_config) if isinstance(quant_config, Fp8Config) else None
        if self.quant_method is not None:
            self.quant_method.create_weights(self)


class RMSNormWeight(nn.Module):

    def __init__(self, hidden_size: int, dtype: torch.dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)


class LlamaAttentionWeights(nn.Module):

    def __init__(self, hidden_size, num_heads, num_kv_heads, head_dim, quant_config, dtype, kv_cache_dtype):
        super().__init__()
        self.qkv_proj = QKVParallelLinear(hidden_size, head_dim, num_heads, num_kv_heads, quant_config, dtype)
        self.o_proj = RowParallelLinear(num_heads * head_dim, hidden_size, quant_config, dtype)
        self.attn = AttentionState(kv_cache_dtype, quant_config)


class LlamaMLPWeights(nn.Module):

    def __init__(self, hidden_size, intermediate_size, quant_config, dtype):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, quant_config, dtype)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, quant_config, dtype)


class LlamaDecoderLayerWeights(nn.Module):

    def __init__(self, hidden_size, intermediate_size, num_heads, num_kv_heads, head_dim, quant_config, dtype,
                 kv_cache_dtype):
        super().__init__()
        self.self_attn = LlamaAttentionWeights(hidden_size, num_heads, num_kv_heads, head_dim, quant_config, dtype,
                                               kv_cache_dtype)
        self.mlp = LlamaMLPWeights(hidden_size, intermediate_size, quant_config, dtype)
        self.input_layernorm = RMSNormWeight(hidden_size, dtype)
        self.post_attention_layernorm = RMSNormWeight(hidden_size, dtype)


class _Model(nn.Module):

    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class LlamaDecoderStack(nn.Module):
    STACKED = [  # (param_name, shard_name, shard_id) - llama.py:434-441
        (".qkv_proj", ".q_proj", "q"),
        (".qkv_proj", ".k_proj", "k"),
        (".qkv_proj", ".v_proj", "v"),
        (".gate_up_proj", ".gate_proj", 0),
        (".gate_up_proj", ".up_proj", 1),
    ]

    def __init__(self, hidden_size: int, intermediate_size: int, num_heads: int, num_kv_heads: int, num_layers: int,
                 quant_config: QuantizationConfig, dtype: torch.dtype = torch.float16, kv_cache_dtype: str = "auto",
                 model_type: Optional[str] = "llama"):
        super().__init__()
        head_dim = hidden_size // num_heads
        self.num_layers, self.model_type = num_layers, model_type
        self.model = _Model([
            LlamaDecoderLayerWeights(hidden_size, intermediate_size, num_heads, num_kv_heads, head_dim, quant_config, dtype,
                                     kv_cache_dtype) for _ in range(num_layers)
        ])

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> int:
        """Feeds checkpoint tensors to the parameters' weight loaders; tensors of modules that are not part of the hot
        path (embeddings, lm_head, rotary caches) are skipped. Returns the number of tensors consumed."""
        params = dict(self.named_parameters())
        used = 0
        for name, loaded in weights:
            if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or "rotary_emb.sin_cached" in name:
                continue
            for param_name, weight_name, shard_id in self.STACKED:
                if weight_name not in name:
                    continue
                name = name.replace(weight_name, param_name)
                if name.endswith(".bias") and name not in params:  # extra bias of GPTQ checkpoints
                    break
                param = params.get(name)
                if param is not None:
                    param.weight_loader(param, loaded, shard_id)
                    used += 1
                break
            else:
                if name.endswith(".bias") and name not in params:
                    continue
                if name.endswith("kv_scale"):  # fp8 checkpoints: ...self_attn.kv_scale -> ...self_attn.attn.kv_scale
                    name = name.replace(".kv_scale", ".attn.kv_scale")
                param = params.get(name)
                if param is None:
                    continue
                getattr(param, "weight_loader", default_weight_loader)(param, loaded)
                used += 1
        return used

    def process_weights_after_loading(self) -> None:
        """Runs every quantization method's post-load step (fp8 scale collapsing, kv_scale extraction ...)."""
        for module in self.modules():
            qm = getattr(module, "quant_method", None)
            if qm is not None and hasattr(qm, "process_weights_after_loading"):
                qm.process_weights_after_loading(module)

    def load_kv_cache_scales(self, quantization_param_path: str) -> None:
        """fp8 KV-cache scaling factors from a JSON file (llama.py:495-519). On ROCm the reference doubles the loaded
        factor (the file holds amax / fp8_max for the e4m3fnuz range, half the e4m3fn one); gfx950 uses OCP e4m3fn, whose
        maximum is 448 like the CUDA build's, so the factor is taken as stored."""
        tp_size, tp_rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        for layer_idx, scaling_factor in kv_cache_scales_loader(quantization_param_path, tp_rank, tp_size, self.num_layers,
                                                                self.model_type):
            self.model.layers[layer_idx].self_attn.attn._kv_scale = float(scaling_factor)
