"""Checkpoint -> device-layout pipeline of the hot path: safetensors iteration, quantization
config discovery, fp8 KV-cache scale files, and the Llama decoder-layer weight mapping (q/k/v -> qkv_proj,
gate/up -> gate_up_proj) on top of the TP-sharded quantized linear layers."""
from neuralmagic_vllm_amd.model_loader.llama import LlamaDecoderStack
from neuralmagic_vllm_amd.model_loader.weight_utils import (default_weight_loader, get_quant_config, kv_cache_scales_loader,
                                                            safetensors_weights_iterator)

__all__ = ["LlamaDecoderStack", "default_weight_loader", "get_quant_config", "kv_cache_scales_loader",
           "safetensors_weights_iterator"]
