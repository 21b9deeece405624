"""TEST INFRASTRUCTURE ONLY — Python face of the CPU oracle (``oracle/liboracle.so``).

The functions here take **CPU** torch tensors and mirror the argument order of the reference's
``vllm/_custom_ops.py`` wrappers so that parity tests read like the reference's own tests.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package. The product package (``neuralmagic_vllm_amd``) must never import it: a product path that
routes through the oracle voids every parity claim.
"""
import ctypes
import os
import subprocess
from typing import List, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def build(force: bool = False) -> str:
    """Compile liboracle.so with g++ (seconds). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_attn_cache.cpp", "oracle_quant.cpp", "oracle_elementwise.cpp",
                                             "numfmt.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu", "oracle works on CPU tensors only"
    return ctypes.c_void_p(t.data_ptr())


def kv_code(kv_cache_dtype: str) -> int:
    if kv_cache_dtype == "auto":
        return 0
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return 1
    if kv_cache_dtype == "fp8_e5m2":
        return 2
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float


# ---------------------------------------------------------------------------------------------
# attention / cache (reference: csrc/attention/attention_kernels.cu, csrc/cache_kernels.cu)
# ---------------------------------------------------------------------------------------------
def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                       block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0,
                       blocksparse_local_blocks=0, blocksparse_vert_stride=0, blocksparse_block_size=64,
                       blocksparse_head_sliding_step=0) -> None:
    S, H, D = query.shape
    assert out.is_contiguous() and block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    lib().orc_paged_attention_v1(
        _p(out), _p(query), _p(key_cache), _p(value_cache), c_int(S), c_int(H), c_int(num_kv_heads), c_int(D),
        c_int(block_size), c_i64(query.stride(0)), c_i64(key_cache.stride(0)), c_i64(key_cache.stride(1)),
        c_f(scale), _p(block_tables), c_int(block_tables.shape[1]), _p(seq_lens), _p(alibi_slopes),
        c_int(DT[query.dtype]), c_int(kv_code(kv_cache_dtype)), c_f(kv_scale), c_int(tp_rank),
        c_int(blocksparse_local_blocks), c_int(blocksparse_vert_stride), c_int(blocksparse_block_size),
        c_int(blocksparse_head_sliding_step))


def paged_attention_v2(out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads, scale,
                       block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                       tp_rank=0, blocksparse_local_blocks=0, blocksparse_vert_stride=0,
                       blocksparse_block_size=64, blocksparse_head_sliding_step=0) -> None:
    S, H, D = query.shape
    lib().orc_paged_attention_v2(
        _p(out), _p(exp_sum), _p(max_logits), _p(tmp_out), _p(query), _p(key_cache), _p(value_cache), c_int(S),
        c_int(H), c_int(num_kv_heads), c_int(D), c_int(block_size), c_i64(query.stride(0)),
        c_i64(key_cache.stride(0)), c_i64(key_cache.stride(1)), c_f(scale), _p(block_tables),
        c_int(block_tables.shape[1]), _p(seq_lens), c_int(max_seq_len), _p(alibi_slopes), c_int(DT[query.dtype]),
        c_int(kv_code(kv_cache_dtype)), c_f(kv_scale), c_int(tp_rank), c_int(blocksparse_local_blocks),
        c_int(blocksparse_vert_stride), c_int(blocksparse_block_size), c_int(blocksparse_head_sliding_step))


def context_attention_fwd(q, k, v, o, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, b_ctx_len, max_input_len,
                          alibi_slopes=None, sliding_window=None) -> None:
    """vllm/attention/ops/prefix_prefill.py:674-812 (same argument order)."""
    i32 = lambda t: t.to(torch.int32).contiguous()
    b_loc_, st_, sl_, cl_ = i32(b_loc), i32(b_start_loc), i32(b_seq_len), i32(b_ctx_len)
    D = q.shape[-1]
    lib().orc_context_attention_fwd(
        _p(o), _p(q), _p(k), _p(v), _p(k_cache), _p(v_cache), _p(b_loc_), _p(st_), _p(sl_), _p(cl_), _p(alibi_slopes),
        c_int(sl_.shape[0]), c_int(q.shape[1]), c_int(k.shape[1]), c_int(D), c_int(v_cache.shape[3]), c_int(k_cache.shape[4]),
        c_i64(q.stride(0)), c_i64(q.stride(1)), c_i64(k.stride(0)), c_i64(k.stride(1)), c_i64(v.stride(0)),
        c_i64(v.stride(1)), c_i64(o.stride(0)), c_i64(o.stride(1)), c_i64(k_cache.stride(0)), c_i64(k_cache.stride(1)),
        c_i64(v_cache.stride(0)), c_i64(v_cache.stride(1)), c_i64(b_loc_.stride(0)),
        c_int(sliding_window if sliding_window and sliding_window > 0 else 0), c_f(1.0 / (D**0.5)), c_int(DT[q.dtype]))


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, kv_scale) -> None:
    T, H, D = key.shape
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    assert slot_mapping.dtype == torch.int64
    lib().orc_reshape_and_cache(_p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), c_int(T),
                                c_int(H), c_int(D), c_int(block_size), c_int(x), c_i64(key.stride(0)),
                                c_i64(value.stride(0)), c_int(DT[key.dtype]), c_int(kv_code(kv_cache_dtype)),
                                c_f(kv_scale))


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype) -> None:
    if kv_cache_dtype != "auto":
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    T, H, D = key.shape
    lib().orc_reshape_and_cache_flash(_p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping),
                                      c_int(T), c_int(H), c_int(D), c_int(key_cache.shape[1]),
                                      c_i64(key_cache.stride(0)), c_i64(key.stride(0)), c_i64(value.stride(0)),
                                      c_int(key.element_size()))


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor], block_mapping) -> None:
    n = len(key_caches)
    if n == 0:
        return
    kp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in key_caches])
    vp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in value_caches])
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size()
    lib().orc_copy_blocks(kp, vp, c_int(n), _p(bm), c_int(bm.shape[0]), c_i64(block_bytes))


def swap_blocks(src, dst, block_mapping) -> None:
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = src[0].numel() * src.element_size()
    lib().orc_swap_blocks(_p(src), _p(dst), _p(bm), c_int(bm.shape[0]), c_i64(block_bytes))


def convert_fp8(output, input, scale: float = 1.0, kv_dtype: str = "fp8") -> None:
    kv = kv_code(kv_dtype) or 1  # "auto" converts as e4m3 (csrc/cache_kernels.cu:353-366)
    if output.dtype == torch.uint8:
        lib().orc_convert_fp8(_p(output), _p(input), c_i64(input.numel()), c_f(scale), c_int(DT[input.dtype]),
                              c_int(kv), c_int(1))
    else:
        lib().orc_convert_fp8(_p(output), _p(input), c_i64(input.numel()), c_f(scale), c_int(DT[output.dtype]),
                              c_int(kv), c_int(0))


# ---------------------------------------------------------------------------------------------
# quantized linear
# ---------------------------------------------------------------------------------------------
def matmul(a, w, out_dtype=None):
    M, K = a.shape
    K2, N = w.shape
    assert K == K2
    out_dtype = out_dtype or a.dtype
    out = torch.empty((M, N), dtype=out_dtype)
    lib().orc_matmul(_p(out), _p(a.contiguous()), _p(w.contiguous()), c_int(M), c_int(N), c_int(K),
                     c_int(DT[a.dtype]), c_int(DT[w.dtype]), c_int(DT[out_dtype]))
    return out


def marlin_unpack(b_q_weight, size_k, size_n, num_bits):
    q = torch.empty((size_k, size_n), dtype=torch.uint8)
    lib().orc_marlin_unpack(_p(b_q_weight.contiguous()), c_int(size_k), c_int(size_n), c_int(num_bits), _p(q))
    return q


def gptq_marlin_dequant(b_q_weight, b_scales, g_idx, num_bits, size_k, size_n, is_k_full=True):
    has_act = g_idx is not None and g_idx.numel() > 0
    w = torch.empty((size_k, size_n), dtype=b_scales.dtype)
    lib().orc_gptq_marlin_dequant(_p(w), _p(b_q_weight.contiguous()), _p(b_scales.contiguous()),
                                  _p(g_idx.to(torch.int32).contiguous() if has_act else None), c_int(size_k),
                                  c_int(size_n), c_int(num_bits), c_int(b_scales.shape[0]), c_int(int(has_act)),
                                  c_int(int(is_k_full)), c_int(DT[b_scales.dtype]))
    return w


def gptq_marlin_gemm(a, b_q_weight, b_scales, g_idx, perm, workspace, num_bits, size_m, size_n, size_k, is_k_full):
    has_act = g_idx is not None and g_idx.numel() > 0
    c = torch.empty((size_m, size_n), dtype=a.dtype)
    lib().orc_gptq_marlin_gemm(_p(c), _p(a.contiguous()), _p(b_q_weight.contiguous()), _p(b_scales.contiguous()),
                               _p(g_idx.to(torch.int32).contiguous() if has_act else None),
                               _p(perm.to(torch.int32).contiguous() if has_act else None), c_int(size_m),
                               c_int(size_n), c_int(size_k), c_int(num_bits), c_int(b_scales.shape[0]),
                               c_int(int(has_act)), c_int(int(is_k_full)), c_int(DT[a.dtype]))
    return c


def marlin_gemm(a, b_q_weight, b_scales, workspace, size_m, size_n, size_k):
    e = torch.empty(0, dtype=torch.int32)
    return gptq_marlin_gemm(a, b_q_weight, b_scales, e, e, workspace, 4, size_m, size_n, size_k, True)


def gptq_marlin_24_gemm(a, b_q_weight, b_meta, b_scales, workspace, num_bits, size_m, size_n, size_k):
    """marlin_24_cuda_kernel.cu:1017-1125 as arithmetic: decode the compressed 2:4 weight (kept values + CUTLASS
    metadata + scales) to the dense fp16 matrix, then fp32-accumulated matmul, result rounded to a.dtype."""
    from oracle import packing
    gs = -1 if b_scales.shape[0] == 1 else size_k // b_scales.shape[0]
    w = packing.marlin_24_decode(b_q_weight, b_meta, b_scales, num_bits, size_k, size_n, gs)
    return matmul(a, w.to(a.dtype))


def fp8_marlin_gemm(a, b_q_weight, b_scales, workspace, num_bits, size_m, size_n, size_k):
    c = torch.empty((size_m, size_n), dtype=a.dtype)
    lib().orc_fp8_marlin_gemm(_p(c), _p(a.contiguous()), _p(b_q_weight.contiguous()), _p(b_scales.contiguous()),
                              c_int(size_m), c_int(size_n), c_int(size_k), c_int(DT[a.dtype]))
    return c


def awq_dequantize(qweight, scales, zeros, split_k_iters=0, thx=0, thy=0):
    K, NC = qweight.shape
    N = NC * 8
    G = K // scales.shape[0]
    w = torch.empty((K, N), dtype=torch.float16)
    lib().orc_awq_dequantize(_p(w), _p(qweight.contiguous()), _p(scales.contiguous()), _p(zeros.contiguous()),
                             c_int(K), c_int(N), c_int(G))
    return w


def awq_gemm(input, qweight, scales, zeros, split_k_iters):
    """Positional order of the C++ op: (in, kernel, scaling_factors, zeros, split_k) — csrc/ops.h:66-68."""
    M, K = input.shape
    N = qweight.shape[1] * 8
    G = K // scales.shape[0]
    c = torch.empty((M, N), dtype=torch.float16)
    lib().orc_awq_gemm(_p(c), _p(input.contiguous()), _p(qweight.contiguous()), _p(scales.contiguous()),
                       _p(zeros.contiguous()), c_int(M), c_int(N), c_int(K), c_int(G))
    return c


def gptq_dequantize(qweight, qzeros, scales, g_idx, bit):
    K = qweight.shape[0] * 32 // bit
    N = qweight.shape[1]
    w = torch.empty((K, N), dtype=torch.float16)
    gi = g_idx.to(torch.int32).contiguous() if (g_idx is not None and g_idx.numel() > 0) else None
    lib().orc_gptq_dequantize(_p(w), _p(qweight.contiguous()), _p(qzeros.contiguous()), _p(scales.contiguous()),
                              _p(gi), c_int(K), c_int(N), c_int(qzeros.shape[0]), c_int(bit))
    return w


def gptq_gemm(a, qweight, qzeros, scales, g_idx, bit):
    """Oracle for the *unshuffled* checkpoint format; callers un-shuffle exllama weights first."""
    w = gptq_dequantize(qweight, qzeros, scales, g_idx, bit)
    return matmul(a, w)


def scaled_fp8_quant(input, scale=None):
    out = torch.empty(input.shape, dtype=torch.uint8)
    x = input.contiguous()
    if scale is None:
        s = torch.zeros(1, dtype=torch.float32)
        lib().orc_scaled_fp8_quant(_p(out), _p(x), _p(s), c_i64(x.numel()), c_int(DT[x.dtype]), c_int(1))
    else:
        s = scale.to(torch.float32).reshape(1).clone()
        lib().orc_scaled_fp8_quant(_p(out), _p(x), _p(s), c_i64(x.numel()), c_int(DT[x.dtype]), c_int(0))
    return out.view(torch.float8_e4m3fn), s


def scaled_int8_quant(input, scale=None):
    x = input.contiguous()
    hidden = x.shape[-1]
    T = x.numel() // hidden
    out = torch.empty(x.shape, dtype=torch.int8)
    if scale is None:
        s = torch.empty((T, 1), dtype=torch.float32)
        lib().orc_scaled_int8_quant(_p(out), _p(x), _p(s), c_int(T), c_int(hidden), c_int(DT[x.dtype]), c_int(1))
    else:
        s = scale.to(torch.float32).reshape(1).clone()
        lib().orc_scaled_int8_quant(_p(out), _p(x), _p(s), c_int(T), c_int(hidden), c_int(DT[x.dtype]), c_int(0))
    return out, s


def scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    """a [M,K] row-major, b [K,N] column-major (b.t() contiguous) — fp8 (e4m3fn) or int8."""
    M, K = a.shape
    N = b.shape[1]
    bt = b.t().contiguous()
    is_fp8 = a.dtype == torch.float8_e4m3fn
    out = torch.empty((M, N), dtype=out_dtype)
    sa = scale_a.to(torch.float32).contiguous().reshape(-1)
    sb = scale_b.to(torch.float32).contiguous().reshape(-1)
    lib().orc_scaled_mm(_p(out), _p(a.contiguous().view(torch.uint8)), _p(bt.view(torch.uint8)), _p(sa),
                        c_int(int(sa.numel() > 1)), _p(sb), c_int(int(sb.numel() > 1)),
                        _p(bias.to(out_dtype).contiguous() if bias is not None else None), c_int(M), c_int(N),
                        c_int(K), c_int(int(is_fp8)), c_int(DT[out_dtype]))
    return out


# ---------------------------------------------------------------------------------------------
# element-wise neighbours (reference: csrc/layernorm_kernels.cu, pos_encoding_kernels.cu, activation_kernels.cu)
# ---------------------------------------------------------------------------------------------
def rms_norm(out, input, weight, epsilon) -> None:
    hidden = input.shape[-1]
    lib().orc_rms_norm(_p(out), _p(input), _p(None), _p(weight), c_f(epsilon), c_int(input.numel() // hidden),
                       c_int(hidden), c_int(DT[input.dtype]), c_int(0))


def fused_add_rms_norm(input, residual, weight, epsilon) -> None:
    hidden = input.shape[-1]
    lib().orc_rms_norm(_p(input), _p(input), _p(residual), _p(weight), c_f(epsilon), c_int(input.numel() // hidden),
                       c_int(hidden), c_int(DT[input.dtype]), c_int(1))


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, cos_sin_cache_offsets=None) -> None:
    num_tokens = query.numel() // query.shape[-1]
    lib().orc_rotary_embedding(_p(positions), _p(query), _p(key), _p(cos_sin_cache), _p(cos_sin_cache_offsets),
                               c_int(cos_sin_cache.shape[1]), c_i64(query.stride(-2)), c_i64(key.stride(-2)),
                               c_int(num_tokens), c_int(query.shape[-1] // head_size), c_int(key.shape[-1] // head_size),
                               c_int(head_size), c_int(int(is_neox)), c_int(DT[query.dtype]))


_ACT = {"silu": 0, "gelu": 1, "gelu_tanh": 2, "gelu_new": 3, "gelu_fast": 4, "gelu_quick": 5}


def act_and_mul(out, x, kind: str) -> None:
    d = x.shape[-1] // 2
    lib().orc_activation(_p(out), _p(x), c_int(x.numel() // x.shape[-1]), c_int(d), c_int(_ACT[kind]), c_int(1),
                         c_int(DT[x.dtype]))


def activation(out, x, kind: str) -> None:
    d = x.shape[-1]
    lib().orc_activation(_p(out), _p(x), c_int(x.numel() // d), c_int(d), c_int(_ACT[kind]), c_int(0),
                         c_int(DT[x.dtype]))
