"""MI355X (gfx950) implementation of nm-vllm's quantized-linear + paged-attention + KV-cache hot path.

``neuralmagic_vllm_amd._custom_ops`` mirrors ``vllm/_custom_ops.py`` of the reference (same function names,
argument order and error behaviour) on top of the C-ABI library ``libnmx_hip.so`` (hand-written HIP kernels,
``include/nmx.h``). There is no CPU or PyTorch fallback: if the library is missing every op raises.
"""
__version__ = "0.1.0"
