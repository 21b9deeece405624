"""Registers the ops under the reference's names in `torch.ops._C`, `torch.ops._C_cache_ops` and
`torch.ops._C_cuda_utils`, with the schemas of csrc/torch_bindings.cpp:18-259, so that the reference's unmodified
`vllm/_custom_ops.py` (which calls `torch.ops._C.<op>`) lands in this package's HIP kernels. Importing this module is
the replacement for `import vllm._C`.

Dispatch key: CUDA (HIP tensors are `cuda` tensors in PyTorch-ROCm). Implementations are thin Python functions over
the C-ABI; under HIP-graph replay (how the reference runs decode) their host overhead disappears.
`_C_custom_ar` is intentionally NOT registered: the reference compiles it out on ROCm (torch_bindings.cpp:261) and
`CustomAllreduce` must stay disabled so that RCCL carries the traffic."""
from typing import List, Optional

import torch

from neuralmagic_vllm_amd import _custom_ops as ops

_libs = []
_registered = False


def _def(lib: torch.library.Library, schema: str, fn) -> None:
    lib.define(schema)
    name = schema.split("(")[0]
    lib.impl(name, fn, "CUDA")


def register() -> None:
    global _registered
    if _registered:
        return
    _registered = True
    c = torch.library.Library("_C", "DEF")
    cache = torch.library.Library("_C_cache_ops", "DEF")
    utils = torch.library.Library("_C_cuda_utils", "DEF")
    _libs.extend([c, cache, utils])

    attn_tail = ("Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
                 "Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, float kv_scale, "
                 "int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size, "
                 "int blocksparse_head_sliding_step) -> ()")
    _def(c, "paged_attention_v1(Tensor! out, " + attn_tail, ops.paged_attention_v1)
    _def(c, "paged_attention_v2(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out, " + attn_tail,
         ops.paged_attention_v2)
    for name in ("silu_and_mul", "gelu_and_mul", "gelu_tanh_and_mul", "gelu_new", "gelu_fast", "gelu_quick"):
        _def(c, f"{name}(Tensor! out, Tensor input) -> ()", getattr(ops, name))
    _def(c, "rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()", ops.rms_norm)
    _def(c, "fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, float epsilon) -> ()", ops.fused_add_rms_norm)
    _def(c, "rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, Tensor cos_sin_cache, "
         "bool is_neox) -> ()", ops.rotary_embedding)
    _def(c, "batched_rotary_embedding(Tensor positions, Tensor! query, Tensor! key, int head_size, Tensor cos_sin_cache, "
         "bool is_neox, int rot_dim, Tensor cos_sin_cache_offsets) -> ()", ops.batched_rotary_embedding)
    _def(c, "awq_dequantize(Tensor kernel, Tensor scaling_factors, Tensor zeros, int split_k_iters, int thx, int thy) -> Tensor",
         ops.awq_dequantize)
    _def(c, "awq_gemm(Tensor in_feats, Tensor kernel, Tensor scaling_factors, Tensor zeros, int split_k_iters) -> Tensor",
         ops.awq_gemm)
    _def(c, "marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, int size_m, int size_n, "
         "int size_k) -> Tensor", ops.marlin_gemm)
    _def(c, "gptq_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor g_idx, Tensor perm, Tensor workspace, "
         "int num_bits, int size_m, int size_n, int size_k, bool is_k_full) -> Tensor", ops.gptq_marlin_gemm)
    _def(c, "gptq_marlin_repack(Tensor b_q_weight, Tensor perm, int size_k, int size_n, int num_bits) -> Tensor",
         ops.gptq_marlin_repack)
    _def(c, "gptq_marlin_24_gemm(Tensor a, Tensor b_q_weight, Tensor b_meta, Tensor b_scales, Tensor workspace, "
         "int num_bits, int size_m, int size_n, int size_k) -> Tensor", ops.gptq_marlin_24_gemm)
    _def(c, "fp8_marlin_gemm(Tensor a, Tensor b_q_weight, Tensor b_scales, Tensor workspace, int num_bits, int size_m, "
         "int size_n, int size_k) -> Tensor", ops.fp8_marlin_gemm)
    _def(c, "gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales, Tensor b_g_idx, "
         "bool use_exllama, int bit) -> Tensor", ops.gptq_gemm)
    _def(c, "gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()", ops.gptq_shuffle)

    # the `Tensor!` operands are written in place by the kernels (no temporary, no copy launch; capturable)
    def _scaled_mm(out, a, b, a_scales, b_scales, bias=None):
        ops.cutlass_scaled_mm(a, b, a_scales, b_scales, out.dtype, bias, out=out)

    _def(c, "cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()",
         _scaled_mm)
    c.define("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool")
    c.impl("cutlass_scaled_mm_supports_fp8", ops.cutlass_scaled_mm_supports_fp8, "CompositeExplicitAutograd")

    def _static_fp8(out, input, scale):
        ops.scaled_fp8_quant(input, scale, out=out)

    def _dynamic_fp8(out, input, scale):
        ops.scaled_fp8_quant(input, None, out=out, dynamic_scale_out=scale)

    _def(c, "static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()", _static_fp8)
    _def(c, "dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()", _dynamic_fp8)

    def _static_i8(out, input, scale):
        ops.scaled_int8_quant(input, scale, out=out)

    def _dynamic_i8(out, input, scale):
        ops.scaled_int8_quant(input, None, out=out, dynamic_scale_out=scale)

    _def(c, "static_scaled_int8_quant(Tensor! out, Tensor input, Tensor scale) -> ()", _static_i8)
    _def(c, "dynamic_scaled_int8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()", _dynamic_i8)

    _def(cache, "swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()", ops.swap_blocks)
    cache.impl("swap_blocks", ops.swap_blocks, "CPU")  # host-side source tensors
    _def(cache, "copy_blocks(Tensor[]! key_caches, Tensor[]! value_caches, Tensor block_mapping) -> ()", ops.copy_blocks)
    _def(cache, "reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, "
         "str kv_cache_dtype, float kv_scale) -> ()", ops.reshape_and_cache)
    _def(cache, "reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
         "Tensor slot_mapping, str kv_cache_dtype) -> ()", ops.reshape_and_cache_flash)
    _def(cache, "convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()", ops.convert_fp8)

    utils.define("get_device_attribute(int attribute, int device_id) -> int")
    utils.impl("get_device_attribute", ops.get_device_attribute, "CompositeExplicitAutograd")
    utils.define("get_max_shared_memory_per_block_device_attribute(int device_id) -> int")
    utils.impl("get_max_shared_memory_per_block_device_attribute", ops.get_max_shared_memory_per_block_device_attribute,
               "CompositeExplicitAutograd")


register()
