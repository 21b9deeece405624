"""Drop-in mirror of the reference's ``vllm/_custom_ops.py`` for the hot path, backed by ``libnmx_hip.so``.

Same function names, argument order and error behaviour (RuntimeError on argument violations) as
``vllm/_custom_ops.py:73-421`` of the reference. Each wrapper only extracts raw device pointers / sizes /
strides and the current HIP stream, then calls the C-ABI (``include/nmx.h``); nothing here computes on the
host, and there is no fallback path.
"""
import ctypes
import os
from typing import List, Optional, Tuple, Type

import torch

from . import _lib

c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_vp = ctypes.c_void_p

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported dtype: {t.dtype}") from None


def _kv(kv_cache_dtype: str) -> int:
    # csrc/quantization/fp8/nvidia/quant_utils.cuh:529-566 (DISPATCH_BY_KV_CACHE_DTYPE)
    if kv_cache_dtype == "auto":
        return 0
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return 1
    if kv_cache_dtype == "fp8_e5m2":
        return 2
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def _p(t: Optional[torch.Tensor]) -> c_vp:
    return c_vp(0) if t is None else c_vp(t.data_ptr())


def _dev(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("expected a GPU (HIP) tensor")


def _stream(t: torch.Tensor) -> c_vp:
    # One process per GPU: the library launches on the CURRENT HIP device (kernel attributes, error state, scratch are
    # per device). The reference wraps every op in an OptionalCUDAGuard (e.g. gptq_marlin.cu:1745); here a tensor on
    # another device is refused loudly instead of being launched against the wrong device's state.
    # (the raw getters: torch.cuda.current_stream() builds a Stream object per call - ~3 of the ~5 us an op call costs the
    # host in eager mode)
    idx = t.device.index
    cur = _raw_device()
    if idx is not None and idx != cur:
        raise RuntimeError(f"tensor on cuda:{idx} but the current device is cuda:{cur}: "
                           "wrap the call in torch.cuda.device(tensor.device)")
    return c_vp(_raw_stream(cur))


_raw_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


def is_custom_op_supported(op_name: str) -> bool:
    name = op_name.split("::")[-1]
    return name in globals() and callable(globals()[name])


# ---------------------------------------------------------------------------------------------------------
# scratch for split-K partial sums
# ---------------------------------------------------------------------------------------------------------
# Round 3: the scratch of a call is a plain torch allocation of exactly the bytes the dispatch can use
# (nmx_*_scratch_bytes; 0 for launches without a K split), made on the calling stream and dropped when the last user drops
# it: the GEMM + reduce launches of a plain op, or the DeferredGemm whose `partial` is a view of it. torch's caching allocator
# hands a freed block out again only to later work of the SAME stream, and inside torch.cuda.graph capture the block comes
# from the graph's private pool and lives as long as the graph - so nothing is pinned per (device, stream) any more (rounds
# 1-2 kept >= 64 MiB per stream forever and let a capture borrow another stream's buffer), two streams never share slabs,
# and a DeferredGemm can no longer be invalidated by a later GEMM on its stream: it owns its slabs.
def _get_scratch(device: torch.device, nbytes: int) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 0), dtype=torch.uint8, device=device)


def reserve_scratch(device, nbytes: int = 0) -> None:
    """Kept for callers of rounds 1-2 (nothing to reserve: the scratch is allocated per call, see above)."""


def release_scratch(device=None) -> None:
    """Returns the cached, currently unused blocks of torch's allocator (where freed scratch lives) to the device."""
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------
# activation / norm / rotary ops (vllm/_custom_ops.py:48-163)
# ---------------------------------------------------------------------------------------------------------
_ACT = {"silu": 0, "gelu": 1, "gelu_tanh": 2, "gelu_new": 3, "gelu_fast": 4, "gelu_quick": 5}


def _gated(out: torch.Tensor, x: torch.Tensor, kind: str) -> None:
    _dev(x)
    d = x.shape[-1] // 2
    if not (x.is_contiguous() and out.is_contiguous()) or out.shape[-1] != d:
        raise RuntimeError("act_and_mul: out [..., d] and input [..., 2 * d] must be contiguous")
    _lib.check(_lib.lib().nmx_act_and_mul(_p(out), _p(x), c_int(x.numel() // x.shape[-1]), c_int(d), c_int(_ACT[kind]),
                                          c_int(_dt(x)), _stream(x)))


def _plain_act(out: torch.Tensor, x: torch.Tensor, kind: str) -> None:
    _dev(x)
    d = x.shape[-1]
    if not (x.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("activation: out and input must be contiguous")
    _lib.check(_lib.lib().nmx_activation(_p(out), _p(x), c_int(x.numel() // d), c_int(d), c_int(_ACT[kind]),
                                         c_int(_dt(x)), _stream(x)))


def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, "silu")


def gelu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, "gelu")


def gelu_tanh_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, "gelu_tanh")


def gelu_fast(out: torch.Tensor, x: torch.Tensor) -> None:
    _plain_act(out, x, "gelu_fast")


def gelu_new(out: torch.Tensor, x: torch.Tensor) -> None:
    _plain_act(out, x, "gelu_new")


def gelu_quick(out: torch.Tensor, x: torch.Tensor) -> None:
    _plain_act(out, x, "gelu_quick")


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    _dev(query)
    num_tokens = query.numel() // query.shape[-1]
    _lib.check(_lib.lib().nmx_rotary_embedding(
        _p(positions), _p(query), _p(key), _p(cos_sin_cache), c_vp(0), c_int(cos_sin_cache.shape[1]),
        c_i64(query.stride(-2)), c_i64(key.stride(-2)), c_int(num_tokens), c_int(query.shape[-1] // head_size),
        c_int(key.shape[-1] // head_size), c_int(head_size), c_int(int(is_neox)), c_int(_dt(query)), _stream(query)))


def batched_rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                             cos_sin_cache: torch.Tensor, is_neox: bool, rot_dim: int,
                             cos_sin_cache_offsets: torch.Tensor) -> None:
    _dev(query)
    num_tokens = cos_sin_cache_offsets.shape[0]
    _lib.check(_lib.lib().nmx_rotary_embedding(
        _p(positions), _p(query), _p(key), _p(cos_sin_cache), _p(cos_sin_cache_offsets), c_int(rot_dim),
        c_i64(query.stride(-2)), c_i64(key.stride(-2)), c_int(num_tokens), c_int(query.shape[-1] // head_size),
        c_int(key.shape[-1] // head_size), c_int(head_size), c_int(int(is_neox)), c_int(_dt(query)), _stream(query)))


def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    _dev(input)
    hidden = input.shape[-1]
    if not (input.is_contiguous() and out.is_contiguous() and weight.is_contiguous()):
        raise RuntimeError("rms_norm: tensors must be contiguous")
    _lib.check(_lib.lib().nmx_rms_norm(_p(out), _p(input), _p(weight), c_f(epsilon), c_int(input.numel() // hidden),
                                       c_int(hidden), c_int(_dt(input)), _stream(input)))


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    _dev(input)
    hidden = input.shape[-1]
    if not (input.is_contiguous() and residual.is_contiguous() and weight.is_contiguous()):
        raise RuntimeError("fused_add_rms_norm: tensors must be contiguous")
    _lib.check(_lib.lib().nmx_fused_add_rms_norm(_p(input), _p(residual), _p(weight), c_f(epsilon),
                                                 c_int(input.numel() // hidden), c_int(hidden), c_int(_dt(input)),
                                                 _stream(input)))


# ---------------------------------------------------------------------------------------------------------
# paged attention (vllm/_custom_ops.py:73-129)
# ---------------------------------------------------------------------------------------------------------
def _attn_common(fn, head_args, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                 block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                 blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                 blocksparse_head_sliding_step, tail_args=()):
    _dev(query)
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("block_tables and seq_lens must be int32")
    num_seqs, num_heads, head_size = query.shape
    if query.stride(1) != head_size or query.stride(2) != 1:
        raise RuntimeError("query must be contiguous in its last two dimensions")
    _lib.check(
        fn(*head_args, _p(query), _p(key_cache), _p(value_cache), c_int(num_seqs), c_int(num_heads),
           c_int(num_kv_heads), c_int(head_size), c_int(block_size), c_i64(query.stride(0)),
           c_i64(key_cache.stride(0)), c_i64(key_cache.stride(1)), c_f(scale), _p(block_tables),
           c_int(block_tables.size(1)), _p(seq_lens), c_int(max_seq_len), _p(alibi_slopes), c_int(_dt(query)),
           c_int(_kv(kv_cache_dtype)), c_f(kv_scale), c_int(tp_rank), c_int(blocksparse_local_blocks),
           c_int(blocksparse_vert_stride), c_int(blocksparse_block_size), c_int(blocksparse_head_sliding_step),
           *tail_args, _stream(query)))


def _v2_fine_partition(query, num_kv_heads, max_seq_len):
    """Partition size the v2 ops run with: the contract's 512 tokens, or - at small batch, where 512-token partitions leave
    most CUs idle - what nmx_paged_attention_partition_size() answers. With a finer split the op works on temporaries of its
    own (the caller's exp_sums / max_logits / tmp_out are sized for 512-token partitions and hold nothing a caller may read
    afterwards: vllm/attention/ops/paged_attn.py:148-158 allocates them per call)."""
    ps = _lib.lib().nmx_paged_attention_partition_size(c_int(query.shape[0]), c_int(query.shape[1]), c_int(num_kv_heads),
                                                       c_int(max_seq_len))
    if ps >= 512:
        return 512, None
    num_seqs, num_heads, head_size = query.shape
    parts = (max_seq_len + ps - 1) // ps
    tmp = torch.empty(num_seqs, num_heads, parts, head_size, dtype=query.dtype, device=query.device)
    sums = torch.empty(2, num_seqs, num_heads, parts, dtype=torch.float32, device=query.device)
    return ps, (sums[0], sums[1], tmp, _v2_counters(query, num_kv_heads))


_V2_COUNTERS = {}  # (device index, stream handle) -> int32 zeros; launches on one stream run one after the other


def _v2_counters(query, num_kv_heads):
    """Arrival counters of the in-kernel v2 reduce: all zero between launches (the kernel puts them back), one tensor per
    (device, stream) so that two launches in flight never share one. OFF unless NMX_ATTN_INKERNEL_REDUCE=1 (environment, read
    per call by this wrapper only): measured level with the reduce launch it saves (batch 1: 9.2 vs 9.6 us per attention call,
    batch 4 / 8 equal - the drain + ticket + agent-scope loads cost what the launch boundary costs), and its cross-workgroup
    hand-off rests on measured, not architecturally guaranteed, cache behaviour (MI355X_MICROARCH.md, valid forms)."""
    if os.environ.get("NMX_ATTN_INKERNEL_REDUCE", "0") != "1" or query.dtype == torch.float32:
        return None
    n = int(_lib.lib().nmx_paged_attention_counters_numel(c_int(query.shape[0]), c_int(query.shape[1]), c_int(num_kv_heads)))
    key = (query.device.index, int(_raw_stream(_raw_device())))
    buf = _V2_COUNTERS.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 1024), dtype=torch.int32, device=query.device)
        _V2_COUNTERS[key] = buf
    return buf


def paged_attention_v1(
    out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    _attn_common(_lib.lib().nmx_paged_attention_v1, (_p(out), ), query, key_cache, value_cache, num_kv_heads, scale,
                 block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                 blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                 blocksparse_head_sliding_step)


def paged_attention_v2(
    out: torch.Tensor,
    exp_sum: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    ps, own = _v2_fine_partition(query, num_kv_heads, max_seq_len)
    if own is not None:
        _attn_common(_lib.lib().nmx_paged_attention_v2_ps, (_p(out), _p(None), _p(own[0]), _p(own[1]), _p(own[2])), query, key_cache,
                     value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                     kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                     blocksparse_head_sliding_step, tail_args=(c_int(ps), _p(own[3])))
        return
    _attn_common(_lib.lib().nmx_paged_attention_v2, (_p(out), _p(exp_sum), _p(max_logits), _p(tmp_out)), query,
                 key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                 alibi_slopes, kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                 blocksparse_block_size, blocksparse_head_sliding_step)


def paged_attention_v1_absmax(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size,
                              max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0, blocksparse_local_blocks=0,
                              blocksparse_vert_stride=0, blocksparse_block_size=64, blocksparse_head_sliding_step=0) -> torch.Tensor:
    """paged_attention_v1 that also returns partial maxima of |out| (float32; their maximum is out.abs().max() exactly) for
    scaled_fp8_quant_partials: the fp8 W8A8 step then has no absmax pass over the attention output."""
    n = _lib.lib().nmx_paged_attention_absmax_numel(c_int(query.shape[0]), c_int(query.shape[1]), c_int(num_kv_heads), c_int(0))
    amax = torch.empty(n, dtype=torch.float32, device=query.device)
    _attn_common(_lib.lib().nmx_paged_attention_v1_absmax, (_p(out), _p(amax)), query, key_cache, value_cache, num_kv_heads,
                 scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                 blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step)
    return amax


def paged_attention_v2_absmax(out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                              seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0,
                              blocksparse_local_blocks=0, blocksparse_vert_stride=0, blocksparse_block_size=64,
                              blocksparse_head_sliding_step=0) -> torch.Tensor:
    """paged_attention_v2 with the same by-product (one maximum per head and sequence, written by the reduce kernel)."""
    n = _lib.lib().nmx_paged_attention_absmax_numel(c_int(query.shape[0]), c_int(query.shape[1]), c_int(num_kv_heads), c_int(1))
    amax = torch.empty(n, dtype=torch.float32, device=query.device)
    ps, own = _v2_fine_partition(query, num_kv_heads, max_seq_len)
    if own is not None:
        _attn_common(_lib.lib().nmx_paged_attention_v2_ps, (_p(out), _p(amax), _p(own[0]), _p(own[1]), _p(own[2])), query, key_cache,
                     value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                     kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                     blocksparse_head_sliding_step, tail_args=(c_int(ps), _p(own[3])))
        return amax
    _attn_common(_lib.lib().nmx_paged_attention_v2_absmax, (_p(out), _p(amax), _p(exp_sum), _p(max_logits), _p(tmp_out)), query,
                 key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                 kv_cache_dtype, kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                 blocksparse_head_sliding_step)
    return amax


class AttnPartials:
    """Partition results of paged_attention_v2_partials: exp_sums / max_logits [S, H, P], tmp [S, H, P, D]; the reduce has not
    run. Hand it to paged_attention_gptq_marlin_gemm (o_proj reduces in its prologue) or call materialize()."""
    __slots__ = ("exp_sums", "max_logits", "tmp", "seq_lens", "part", "out")

    def __init__(self, exp_sums, max_logits, tmp, seq_lens, part):
        self.exp_sums, self.max_logits, self.tmp, self.seq_lens, self.part, self.out = exp_sums, max_logits, tmp, seq_lens, part, None

    def materialize(self) -> torch.Tensor:
        """The attention output [S, H, D] (the reduce kernel as a launch of its own)."""
        if self.out is None:
            S, H, P, D = self.tmp.shape
            self.out = torch.empty(S, H, D, dtype=self.tmp.dtype, device=self.tmp.device)
            _lib.check(_lib.lib().nmx_paged_attention_v2_reduce(_p(self.out), _p(self.exp_sums), _p(self.max_logits), _p(self.tmp),
                                                                _p(self.seq_lens), c_int(S), c_int(H), c_int(D), c_int(P),
                                                                c_int(self.part), c_int(_dt(self.tmp)), _stream(self.tmp)))
        return self.out


def paged_attention_v2_partials(query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                                alibi_slopes, kv_cache_dtype, kv_scale, tp_rank=0, blocksparse_local_blocks=0,
                                blocksparse_vert_stride=0, blocksparse_block_size=64, blocksparse_head_sliding_step=0) -> AttnPartials:
    """paged_attention_v2's partition launch alone (partition size: the library's choice, see _v2_fine_partition)."""
    ps = _lib.lib().nmx_paged_attention_partition_size(c_int(query.shape[0]), c_int(query.shape[1]), c_int(num_kv_heads),
                                                       c_int(max_seq_len))
    num_seqs, num_heads, head_size = query.shape
    parts = (max_seq_len + ps - 1) // ps
    tmp = torch.empty(num_seqs, num_heads, parts, head_size, dtype=query.dtype, device=query.device)
    sums = torch.empty(2, num_seqs, num_heads, parts, dtype=torch.float32, device=query.device)
    _attn_common(_lib.lib().nmx_paged_attention_v2_partials, (_p(sums[0]), _p(sums[1]), _p(tmp)), query, key_cache, value_cache,
                 num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                 blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step,
                 tail_args=(c_int(ps), ))
    return AttnPartials(sums[0], sums[1], tmp, seq_lens, ps)


def paged_attention_gptq_marlin_gemm(partials: AttnPartials, b_q_weight: torch.Tensor, b_scales: torch.Tensor, g_idx: torch.Tensor,
                                     perm: torch.Tensor, workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int, size_k: int,
                                     is_k_full: bool) -> "DeferredGemm":
    """o_proj on the attention output of paged_attention_v2_partials (LlamaAttention.forward, models/llama.py:171-172), as
    gptq_marlin_gemm_deferred returns it. At batch <= 16 (head size 128, int4 without act-order, the decode kernel's 4-wave
    shape: nmx_gptq_marlin_gemm_attn_supported) the GEMM reduces the partitions in its prologue - every wave the heads of its
    own K slice, with the reduce kernel's device function - and no reduce kernel is launched; otherwise reduce launch + GEMM.
    Same bits either way (tests/test_fused_gpu.py::test_attn_reduce_gemm)."""
    S, H, P, D = partials.tmp.shape
    has_idx = g_idx is not None and g_idx.numel() > 0
    fused = (partials.out is None and not has_idx and size_m == S and size_k == H * D and
             _lib.lib().nmx_gptq_marlin_gemm_attn_supported(c_int(size_m), c_int(size_n), c_int(size_k), c_int(b_scales.shape[0]),
                                                            c_int(num_bits), c_int(_dt(partials.tmp)), c_int(H), c_int(D), c_int(P)))
    if not fused:
        a = partials.materialize().view(S, H * D)
        return gptq_marlin_gemm_deferred(a, b_q_weight, b_scales, g_idx, perm, workspace, num_bits, size_m, size_n, size_k, is_k_full)
    t = partials.tmp
    c = torch.empty((size_m, size_n), dtype=t.dtype, device=t.device)
    scratch = _marlin_scratch(t, size_m, size_n, size_k)
    splits = c_int(1)
    _lib.check(_lib.lib().nmx_gptq_marlin_gemm_attn(
        _p(partials.exp_sums), _p(partials.max_logits), _p(t), _p(partials.seq_lens), c_int(partials.part), c_int(P), c_int(H), c_int(D),
        _p(b_q_weight), _p(b_scales), _p(c), c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n),
        c_int(size_k), c_int(num_bits), c_int(b_scales.shape[0]), c_int(_dt(t)), ctypes.byref(splits), _stream(t)))
    if (splits.value & 0xff) > 1:
        partial, n = _slabs(scratch, splits.value, size_m, size_n)
        return DeferredGemm(c, partial, n)
    return DeferredGemm(c, None, 1)


# ---------------------------------------------------------------------------------------------------------
# KV-cache ops (vllm/_custom_ops.py:370-412)
# ---------------------------------------------------------------------------------------------------------
def reshape_and_cache(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    kv_cache_dtype: str,
    kv_scale: float,
) -> None:
    _dev(key)
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    num_tokens, num_heads, head_size = key.shape
    _lib.check(_lib.lib().nmx_reshape_and_cache(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), c_int(num_tokens), c_int(num_heads),
        c_int(head_size), c_int(key_cache.size(3)), c_int(key_cache.size(4)), c_i64(key.stride(0)),
        c_i64(value.stride(0)), c_int(_dt(key)), c_int(_kv(kv_cache_dtype)), c_f(kv_scale), _stream(key)))


def reshape_and_cache_flash(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    kv_cache_dtype: str,
) -> None:
    _dev(key)
    if kv_cache_dtype != "auto":  # csrc/cache_kernels.cu:288-291
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    if key_cache.stride(0) != value_cache.stride(0):
        raise RuntimeError("k_cache.stride(0) == v_cache.stride(0) must hold")
    num_tokens, num_heads, head_size = key.shape
    _lib.check(_lib.lib().nmx_reshape_and_cache_flash(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping), c_int(num_tokens), c_int(num_heads),
        c_int(head_size), c_int(key_cache.size(1)), c_i64(key_cache.stride(0)), c_i64(key.stride(0)),
        c_i64(value.stride(0)), c_int(key.element_size()), _stream(key)))


_ptr_arrays = {}


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    num_layers = len(key_caches)
    if num_layers != len(value_caches):
        raise RuntimeError("key_caches and value_caches must have the same length")
    if num_layers == 0:
        return
    dev = key_caches[0].device
    _dev(key_caches[0])
    # device arrays of base pointers; the reference re-uploads them (with a sync) on every call
    # (csrc/cache_kernels.cu:126-133). KV caches live for the engine's lifetime, so cache the upload.
    key = (dev.index, tuple(t.data_ptr() for t in key_caches), tuple(t.data_ptr() for t in value_caches))
    arrs = _ptr_arrays.get(key)
    if arrs is None:
        kp = torch.tensor([t.data_ptr() for t in key_caches], dtype=torch.int64).to(dev)
        vp = torch.tensor([t.data_ptr() for t in value_caches], dtype=torch.int64).to(dev)
        if len(_ptr_arrays) > 64:
            _ptr_arrays.clear()
        arrs = _ptr_arrays[key] = (kp, vp)
    bm = block_mapping
    if bm.dtype != torch.int64 or not bm.is_cuda or not bm.is_contiguous():
        bm = bm.to(device=dev, dtype=torch.int64).contiguous()
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size()
    _lib.check(_lib.lib().nmx_copy_blocks(_p(arrs[0]), _p(arrs[1]), _p(bm), c_int(num_layers), c_int(bm.size(0)),
                                          c_i64(block_bytes), _stream(key_caches[0])))


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    # csrc/cache_kernels.cu:24-63
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")
        kind = 0
    elif src.is_cuda and not dst.is_cuda:
        kind = 1
    elif not src.is_cuda and dst.is_cuda:
        kind = 2
    else:
        raise RuntimeError("Invalid device combination")
    if block_mapping.is_cuda:
        raise RuntimeError("block_mapping must be on CPU")
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = src.element_size() * src[0].numel()
    gpu = src if src.is_cuda else dst
    with torch.cuda.device(gpu.device):
        _lib.check(_lib.lib().nmx_swap_blocks(_p(src), _p(dst), _p(bm), c_int(bm.size(0)), c_i64(block_bytes),
                                              c_int(kind), _stream(gpu)))


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0, kv_dtype: str = "fp8") -> None:
    # csrc/cache_kernels.cu:339-389
    if not input.is_cuda:
        raise RuntimeError("src must be on a GPU")
    if not output.is_cuda:
        raise RuntimeError("dst must be on a GPU")
    if input.device != output.device:
        raise RuntimeError("src and dst must be on the same GPU")
    kv = _kv(kv_dtype) or 1  # "auto" converts as e4m3 (cache_kernels.cu:353-366)
    if input.numel() != output.numel():
        raise RuntimeError("convert_fp8: size mismatch")
    if output.dtype == torch.uint8 and input.dtype != torch.uint8:
        _lib.check(_lib.lib().nmx_convert_fp8(_p(output), _p(input), c_i64(input.numel()), c_f(scale),
                                              c_int(_dt(input)), c_int(kv), c_int(1), _stream(input)))
    elif input.dtype == torch.uint8:
        _lib.check(_lib.lib().nmx_convert_fp8(_p(output), _p(input), c_i64(input.numel()), c_f(scale),
                                              c_int(_dt(output)), c_int(kv), c_int(0), _stream(input)))
    else:
        raise RuntimeError("convert_fp8: one side must be uint8 (fp8 storage)")


# ---------------------------------------------------------------------------------------------------------
# Marlin family (vllm/_custom_ops.py:200-280)
# ---------------------------------------------------------------------------------------------------------
def gptq_marlin_repack(b_q_weight: torch.Tensor, perm: torch.Tensor, size_k: int, size_n: int,
                       num_bits: int) -> torch.Tensor:
    # csrc/quantization/gptq_marlin/gptq_marlin_repack.cu:276-348
    _dev(b_q_weight)
    if num_bits not in (4, 8):
        raise RuntimeError(f"num_bits must be 4 or 8. Got = {num_bits}")
    pack_factor = 32 // num_bits
    if size_k % 16 != 0:
        raise RuntimeError(f"size_k = {size_k} is not divisible by tile_k_size = 16")
    if (size_k // pack_factor) != b_q_weight.size(0):
        raise RuntimeError(f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.size(0)}, size_k = {size_k}, "
                           f"pack_factor = {pack_factor}")
    if b_q_weight.size(1) != size_n:
        raise RuntimeError(f"b_q_weight.size(1) = {b_q_weight.size(1)} is not size_n = {size_n}")
    if not b_q_weight.is_contiguous():
        raise RuntimeError("b_q_weight is not contiguous")
    if b_q_weight.dtype != torch.int32:
        raise RuntimeError("b_q_weight type is not kInt")
    has_perm = perm.numel() != 0
    if has_perm and (perm.dtype != torch.int32 or not perm.is_cuda or not perm.is_contiguous()):
        raise RuntimeError("perm must be a contiguous int32 GPU tensor")
    out = torch.empty((size_k // 16, size_n * 16 // pack_factor), dtype=torch.int32, device=b_q_weight.device)
    _lib.check(_lib.lib().nmx_gptq_marlin_repack(_p(b_q_weight), _p(perm if has_perm else None), _p(out),
                                                 c_int(size_k), c_int(size_n), c_int(num_bits),
                                                 _stream(b_q_weight)))
    return out


def _marlin_scratch(a: torch.Tensor, size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    need = _lib.lib().nmx_marlin_gemm_scratch_bytes(c_int(size_m), c_int(size_n), c_int(size_k))
    return _get_scratch(a.device, int(need))


def gptq_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor, g_idx: torch.Tensor,
                     perm: torch.Tensor, workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int,
                     size_k: int, is_k_full: bool) -> torch.Tensor:
    # checks mirror csrc/quantization/gptq_marlin/gptq_marlin.cu:1735-1843
    if num_bits not in (4, 8):
        raise RuntimeError(f"num_bits must be 4 or 8. Got = {num_bits}")
    pack_factor = 32 // num_bits
    if a.size(0) != size_m:
        raise RuntimeError(f"Shape mismatch: a.size(0) = {a.size(0)}, size_m = {size_m}")
    if a.size(1) != size_k:
        raise RuntimeError(f"Shape mismatch: a.size(1) = {a.size(1)}, size_k = {size_k}")
    if size_k % 16 != 0:
        raise RuntimeError(f"size_k = {size_k} is not divisible by tile_size = 16")
    if (size_k // 16) != b_q_weight.size(0):
        raise RuntimeError(f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.size(0)}, size_k = {size_k}, "
                           "tile_size = 16")
    if b_q_weight.size(1) % 16 != 0:
        raise RuntimeError(f"b_q_weight.size(1) = {b_q_weight.size(1)} is not divisible by tile_size = 16")
    actual_size_n = (b_q_weight.size(1) // 16) * pack_factor
    if size_n != actual_size_n:
        raise RuntimeError(f"size_n = {size_n}, actual_size_n = {actual_size_n}")
    for name, t in (("A", a), ("b_q_weight", b_q_weight), ("b_scales", b_scales), ("g_idx", g_idx), ("perm", perm)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} is not on GPU")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} is not contiguous")
    if not ((g_idx.size(0) == 0 and perm.size(0) == 0) or (g_idx.size(0) == size_k and perm.size(0) == size_k)):
        raise RuntimeError(f"Unexpected g_idx.size(0) = {g_idx.size(0)} and perm.size(0) = {perm.size(0)}, "
                           f"where size_k = {size_k}")
    if b_scales.dim() != 2:
        raise RuntimeError(f"b_scales rank = {b_scales.dim()} is not 2")
    if b_scales.size(1) != size_n:
        raise RuntimeError(f"b_scales dim 1 = {b_scales.size(1)} is not size_n = {size_n}")
    if a.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("gpt_marlin_gemm only supports bfloat16 and float16")
    if b_scales.dtype != a.dtype:
        raise RuntimeError("b_scales must have the dtype of a")
    has_act = g_idx.size(0) != 0
    if has_act and (g_idx.dtype != torch.int32 or perm.dtype != torch.int32):
        raise RuntimeError("g_idx and perm must be int32")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    _lib.check(_lib.lib().nmx_gptq_marlin_gemm(
        _p(a), _p(b_q_weight), _p(b_scales), _p(g_idx if has_act else None), _p(perm if has_act else None), _p(c),
        c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n), c_int(size_k),
        c_int(num_bits), c_int(b_scales.size(0)), c_int(int(bool(is_k_full))), c_int(_dt(a)), _stream(a)))
    return c


def marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor, workspace: torch.Tensor,
                size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    # checks mirror csrc/quantization/marlin/dense/marlin_cuda_kernel.cu:1045-1100
    if a.size(0) != size_m:
        raise RuntimeError(f"Shape mismatch: a.size(0) = {a.size(0)}, size_m = {size_m}")
    if a.size(1) != size_k:
        raise RuntimeError(f"Shape mismatch: a.size(1) = {a.size(1)}, size_k = {size_k}")
    if size_k % 16 != 0:
        raise RuntimeError(f"size_k = {size_k} is not divisible by tile_size = 16")
    if (size_k // 16) != b_q_weight.size(0):
        raise RuntimeError(f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.size(0)}, size_k = {size_k}, "
                           "tile_size = 16")
    if b_scales.size(1) != size_n:
        raise RuntimeError(f"b_scales.size(1) = {b_scales.size(1)}, size_n = {size_n}")
    if size_k % b_scales.size(0) != 0:
        raise RuntimeError(f"size_k = {size_k}, is not divisible by b_scales.size(0) = {b_scales.size(0)}")
    if a.dtype != torch.float16:
        raise RuntimeError("marlin_gemm only supports float16")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    _lib.check(_lib.lib().nmx_marlin_gemm(_p(a.contiguous()), _p(b_q_weight), _p(b_scales), _p(c),
                                          c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()),
                                          c_int(size_m), c_int(size_n), c_int(size_k), c_int(b_scales.size(0)),
                                          _stream(a)))
    return c


# ---------------------------------------------------------------------------------------------------------
# fused decode path (extensions, not part of vllm._custom_ops): the split-K reduction of a Marlin GEMM is deferred to the
# op that consumes its output. Every combination is bit-identical to the plain op sequence; the plain ops stay available.
# ---------------------------------------------------------------------------------------------------------
_SPLITK_F16 = 0x100  # NMX_SPLITK_F16 (include/nmx.h): the slabs hold fp16 partial sums


def _slabs(scratch: torch.Tensor, coded: int, size_m: int, size_n: int):
    """(partial view, slab count) of what a *_deferred entry reported in *splits_out (count | NMX_SPLITK_F16)."""
    n = coded & 0xff
    if coded & _SPLITK_F16:
        return scratch[:n * size_m * size_n * 2].view(torch.float16).view(n, size_m, size_n), n
    return scratch[:n * size_m * size_n * 4].view(torch.float32).view(n, size_m, size_n), n


class DeferredGemm:
    """Output of gptq_marlin_gemm_deferred: `out` [M, N] (valid iff splits == 1), else `partial` [splits, M, N] - fp32, or fp16
    from the M > 64 Marlin kernels (round 3: half the slab traffic; the consumers and the reduce entry take the dtype with the
    count, see `coded`)."""
    __slots__ = ("out", "partial", "splits", "sa", "sb")

    def __init__(self, out, partial, splits, sa=None, sb=None):
        self.out, self.partial, self.splits = out, partial, splits
        self.sa, self.sb = sa, sb  # per-tensor scales of a deferred fp8 scaled_mm (applied by the consumer), else None

    @property
    def coded(self) -> int:
        """The `splits` argument of the C-ABI consumers: slab count | NMX_SPLITK_F16 for fp16 slabs."""
        return self.splits | (_SPLITK_F16 if self.partial is not None and self.partial.dtype == torch.float16 else 0)

    def materialize(self) -> torch.Tensor:
        """Plain reduction (what the reduce launch would have produced), for consumers without a fused form."""
        if self.splits > 1:
            if self.sa is not None:  # fp8 scaled_mm: the scale epilogue belongs to the reduction (the GEMM's own reduce kernel)
                m, n = self.out.shape
                _lib.check(_lib.lib().nmx_splitk_reduce_scaled(_p(self.out), _p(self.partial), c_int(self.coded), _p(self.sa),
                                                               _p(self.sb), c_int(m), c_int(n), c_i64(self.out.stride(0)),
                                                               c_int(_dt(self.out)), _stream(self.out)))
            else:  # the GEMMs' own reduce kernel: slabs summed in the order s = 0, 1, ... like the plain op
                m, n = self.out.shape
                _lib.check(_lib.lib().nmx_splitk_reduce(_p(self.out), _p(self.partial), c_int(self.coded), c_int(m), c_int(n),
                                                        c_int(_dt(self.out)), _stream(self.out)))
            self.splits = 1
        return self.out


def gptq_marlin_gemm_deferred(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor, g_idx: torch.Tensor,
                              perm: torch.Tensor, workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int,
                              size_k: int, is_k_full: bool) -> DeferredGemm:
    _dev(a)
    if a.dim() != 2 or a.shape[0] != size_m or a.shape[1] != size_k:
        raise RuntimeError(f"Shape mismatch: a.size = {tuple(a.shape)}, size_m = {size_m}, size_k = {size_k}")
    if not a.is_contiguous():
        raise RuntimeError("A is not contiguous")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return DeferredGemm(c, None, 1)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    splits = c_int(1)
    has_idx = g_idx is not None and g_idx.numel() > 0
    _lib.check(_lib.lib().nmx_gptq_marlin_gemm_deferred(
        _p(a), _p(b_q_weight), _p(b_scales), _p(g_idx if has_idx else None), _p(perm if has_idx else None), _p(c),
        c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n), c_int(size_k),
        c_int(num_bits), c_int(b_scales.shape[0]), c_int(int(is_k_full)), c_int(_dt(a)), ctypes.byref(splits), _stream(a)))
    if (splits.value & 0xff) > 1:
        partial, n = _slabs(scratch, splits.value, size_m, size_n)
        return DeferredGemm(c, partial, n)
    return DeferredGemm(c, None, 1)


def gptq_marlin_gemm_silu_and_mul(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor, g_idx: torch.Tensor,
                                  perm: torch.Tensor, workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int,
                                  size_k: int, is_k_full: bool) -> torch.Tensor:
    """silu_and_mul(gptq_marlin_gemm(a, gate_up weight ...)) -> [size_m, size_n / 2] (LlamaMLP's gate_up_proj + SiluAndMul,
    models/llama.py:79-83). One launch where the wide-tile kernel runs without a K split (the activation is its epilogue),
    otherwise GEMM + consumer; bit-identical to the two ops either way (tests/test_fused_gpu.py)."""
    _dev(a)
    if a.dim() != 2 or a.shape[0] != size_m or a.shape[1] != size_k:
        raise RuntimeError(f"Shape mismatch: a.size = {tuple(a.shape)}, size_m = {size_m}, size_k = {size_k}")
    if not a.is_contiguous():
        raise RuntimeError("A is not contiguous")
    if size_n % 16 != 0:
        raise RuntimeError(f"size_n = {size_n} must be a multiple of 16")
    act = torch.empty((size_m, size_n // 2), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return act
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)  # only touched on the two-launch route
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    has_idx = g_idx is not None and g_idx.numel() > 0
    _lib.check(_lib.lib().nmx_gptq_marlin_gemm_silu_and_mul(
        _p(a), _p(b_q_weight), _p(b_scales), _p(g_idx if has_idx else None), _p(perm if has_idx else None), _p(c), _p(act),
        c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n), c_int(size_k),
        c_int(num_bits), c_int(b_scales.shape[0]), c_int(int(is_k_full)), c_int(_dt(a)), _stream(a)))
    return act


def fused_add_rms_norm_gptq_marlin_gemm(g: DeferredGemm, residual: torch.Tensor, norm_weight: torch.Tensor, epsilon: float,
                                        b_q_weight: torch.Tensor, b_scales: torch.Tensor, g_idx: torch.Tensor, perm: torch.Tensor,
                                        workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int, size_k: int,
                                        is_k_full: bool, silu_and_mul: bool = False):
    """fused_add_rms_norm(g, residual) followed by gptq_marlin_gemm[_silu_and_mul] on the normed rows - LlamaDecoderLayer's
    post_attention_layernorm + gate_up_proj + act, or the next layer's input_layernorm + qkv_proj (models/llama.py:205-230).
    Returns (result, residual): result is the DeferredGemm of gptq_marlin_gemm_deferred (silu_and_mul=False) or the activation
    tensor [size_m, size_n / 2]; residual is the tensor that now holds the updated residual stream.
    At ONE row by default (fp16 / bf16, int4 without act-order, the decode kernel's shapes: nmx_gptq_marlin_gemm_norm_supported) this is ONE
    launch - every workgroup of the GEMM computes the norm in its prologue while its first weight loads are in flight, and the
    residual goes to a NEW tensor; otherwise the two ops run one after the other and the residual is updated in place. The
    results are bit-identical either way (tests/test_fused_gpu.py)."""
    _dev(residual)
    has_idx = g_idx is not None and g_idx.numel() > 0
    fused = (g.splits >= 2 and g.sa is None and not has_idx and residual.dtype in (torch.float16, torch.bfloat16) and residual.is_contiguous()
             and tuple(residual.shape) == (size_m, size_k) and _lib.lib().nmx_gptq_marlin_gemm_norm_supported(
                 c_int(size_m), c_int(size_n), c_int(size_k), c_int(b_scales.shape[0]), c_int(num_bits), c_int(_dt(residual)),
                 c_int(int(silu_and_mul))))
    if not fused:
        h = fused_add_rms_norm_splitk(g, residual, norm_weight, epsilon)
        if silu_and_mul:
            return gptq_marlin_gemm_silu_and_mul(h, b_q_weight, b_scales, g_idx, perm, workspace, num_bits, size_m, size_n, size_k,
                                                 is_k_full), residual
        return gptq_marlin_gemm_deferred(h, b_q_weight, b_scales, g_idx, perm, workspace, num_bits, size_m, size_n, size_k,
                                         is_k_full), residual
    res_out = torch.empty_like(residual)
    c = torch.empty((size_m, size_n), dtype=residual.dtype, device=residual.device)
    act = torch.empty((size_m, size_n // 2), dtype=residual.dtype, device=residual.device) if silu_and_mul else None
    scratch = _marlin_scratch(residual, size_m, size_n, size_k)
    splits = c_int(1)
    _lib.check(_lib.lib().nmx_gptq_marlin_gemm_norm(
        _p(g.partial), c_int(g.coded), _p(residual), _p(res_out), _p(norm_weight), c_f(epsilon), _p(b_q_weight), _p(b_scales), _p(c),
        _p(act), c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n), c_int(size_k),
        c_int(num_bits), c_int(b_scales.shape[0]), c_int(_dt(residual)), ctypes.byref(splits), _stream(residual)))
    g.splits, g.out, g.partial = 1, None, None  # consumed: the reduced tensor never existed (a later materialize() returns None, loudly useless)
    if silu_and_mul:
        return act, res_out
    if (splits.value & 0xff) > 1:
        partial, n = _slabs(scratch, splits.value, size_m, size_n)
        return DeferredGemm(c, partial, n), res_out
    return DeferredGemm(c, None, 1), res_out


def fused_add_rms_norm_splitk(g: DeferredGemm, residual: torch.Tensor, weight: torch.Tensor, epsilon: float,
                              want_absmax: bool = False):
    """fused_add_rms_norm(g.out, residual, ...) on the deferred GEMM output; returns the normed tensor (g.out's storage), or
    (normed, absmax [T] float32) with want_absmax (for scaled_fp8_quant_partials)."""
    out = g.out
    if g.splits == 1:
        if want_absmax:
            return out, fused_add_rms_norm_absmax(out, residual, weight, epsilon)
        fused_add_rms_norm(out, residual, weight, epsilon)
        return out
    amax = torch.empty(out.shape[0], dtype=torch.float32, device=out.device) if want_absmax else None
    if g.sa is not None or want_absmax:
        _lib.check(_lib.lib().nmx_fused_add_rms_norm_splitk_scaled(
            _p(out), _p(g.partial), c_int(g.coded), _p(g.sa), _p(g.sb), _p(residual), _p(weight), c_f(epsilon),
            c_int(out.shape[0]), c_int(out.shape[1]), c_int(_dt(out)), _p(amax), _stream(out)))
    else:
        _lib.check(_lib.lib().nmx_fused_add_rms_norm_splitk(_p(out), _p(g.partial), c_int(g.coded), _p(residual), _p(weight),
                                                            c_f(epsilon), c_int(out.shape[0]), c_int(out.shape[1]),
                                                            c_int(_dt(out)), _stream(out)))
    g.splits = 1
    return (out, amax) if want_absmax else out


def silu_and_mul_splitk(out: torch.Tensor, g: DeferredGemm, want_absmax: bool = False):
    """silu_and_mul(out, g.out) on the deferred GEMM output; with want_absmax returns the per-token |max| of out."""
    if g.splits == 1:
        if want_absmax:
            return silu_and_mul_absmax(out, g.out)
        silu_and_mul(out, g.out)
        return None
    T, d = out.shape
    amax = torch.empty(T, dtype=torch.float32, device=out.device) if want_absmax else None
    if g.sa is not None or want_absmax:
        _lib.check(_lib.lib().nmx_silu_and_mul_splitk_scaled(_p(out), _p(g.partial), c_int(g.coded), _p(g.sa), _p(g.sb), c_int(T),
                                                             c_int(d), c_int(_dt(out)), _p(amax), _stream(out)))
    else:
        _lib.check(_lib.lib().nmx_silu_and_mul_splitk(_p(out), _p(g.partial), c_int(g.coded), c_int(T), c_int(d), c_int(_dt(out)),
                                                      _stream(out)))
    return amax


def rope_reshape_and_cache(positions: torch.Tensor, g, num_heads: int, num_kv_heads: int, head_size: int,
                           cos_sin_cache: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                           slot_mapping: torch.Tensor, kv_cache_dtype: str, kv_scale: float) -> torch.Tensor:
    """rotary_embedding (NeoX, whole head) on q / k of the fused qkv row + reshape_and_cache of k / v in one launch.
    g: a DeferredGemm or a plain [T, (H + 2 KVH) * D] tensor. Returns the qkv tensor (q, k rotated)."""
    if not isinstance(g, DeferredGemm):
        g = DeferredGemm(g, None, 1)
    qkv = g.out
    if not qkv.is_contiguous() or qkv.shape[1] != (num_heads + 2 * num_kv_heads) * head_size:
        raise RuntimeError("rope_reshape_and_cache: qkv must be a contiguous [T, (H + 2 KVH) * D] tensor")
    if cos_sin_cache.shape[1] != head_size or cos_sin_cache.dtype != qkv.dtype:
        raise RuntimeError("rope_reshape_and_cache: rotary over the whole head, cache in the activation dtype")
    block_size = value_cache.shape[3]
    if g.splits > 1 and g.sa is not None:
        _lib.check(_lib.lib().nmx_rope_reshape_and_cache_scaled(
            _p(positions), _p(qkv), _p(g.partial), c_int(g.coded), _p(g.sa), _p(g.sb), _p(cos_sin_cache), _p(key_cache),
            _p(value_cache), _p(slot_mapping), c_int(qkv.shape[0]), c_int(num_heads), c_int(num_kv_heads), c_int(head_size),
            c_int(block_size), c_int(_dt(qkv)), c_int(_kv(kv_cache_dtype)), c_f(kv_scale), _stream(qkv)))
        g.splits = 1
        return qkv
    _lib.check(_lib.lib().nmx_rope_reshape_and_cache(
        _p(positions), _p(qkv), _p(g.partial), c_int(g.coded), _p(cos_sin_cache), _p(key_cache), _p(value_cache),
        _p(slot_mapping), c_int(qkv.shape[0]), c_int(num_heads), c_int(num_kv_heads), c_int(head_size), c_int(block_size),
        c_int(_dt(qkv)), c_int(_kv(kv_cache_dtype)), c_f(kv_scale), _stream(qkv)))
    g.splits = 1
    return qkv


def gptq_marlin_24_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_meta: torch.Tensor, b_scales: torch.Tensor,
                        workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    # checks mirror csrc/quantization/marlin/sparse/marlin_24_cuda_kernel.cu:1024-1082
    if num_bits not in (4, 8):
        raise RuntimeError(f"num_bits must be 4 or 8. Got = {num_bits}")
    pack_factor = 32 // num_bits
    if a.size(0) != size_m:
        raise RuntimeError(f"Shape mismatch: a.size(0) = {a.size(0)}, size_m = {size_m}")
    if a.size(1) != size_k:
        raise RuntimeError(f"Shape mismatch: a.size(1) = {a.size(1)}, size_k = {size_k}")
    if size_k % 16 != 0:
        raise RuntimeError(f"size_k = {size_k} is not divisible by tile_size = 16")
    if (size_k // 16 // 2) != b_q_weight.size(0):
        raise RuntimeError(f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.size(0)}, size_k = {size_k}, "
                           "tile_size = 16")
    if b_scales.size(1) != size_n:
        raise RuntimeError(f"b_scales.size(1) = {b_scales.size(1)}, size_n = {size_n}")
    if b_q_weight.size(1) % 16 != 0:
        raise RuntimeError(f"b_q_weight.size(1) = {b_q_weight.size(1)} is not divisible by tile_size = 16")
    actual_size_n = (b_q_weight.size(1) // 16) * pack_factor
    if size_n != actual_size_n:
        raise RuntimeError(f"size_n = {size_n}, actual_size_n = {actual_size_n}")
    if b_meta.size(0) != size_k // 8 // 2 // 2:
        raise RuntimeError(f"b_meta.size(0) = {b_meta.size(0)} is not size_k / 8 / 2 / 2 = {size_k // 8 // 2 // 2}")
    if b_meta.size(1) != size_n * 2:
        raise RuntimeError(f"b_meta.size(1) = {b_meta.size(1)} is not size_n * 2 = {size_n * 2}")
    for name, t in (("A", a), ("b_q_weight", b_q_weight), ("b_meta", b_meta), ("b_scales", b_scales)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} is not on GPU")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} is not contiguous")
    if a.dtype != torch.float16:
        raise RuntimeError("gptq_marlin_24_gemm only supports float16")
    if b_meta.dtype != torch.int16:
        raise RuntimeError("b_meta must be int16")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    _lib.check(_lib.lib().nmx_gptq_marlin_24_gemm(_p(a), _p(b_q_weight), _p(b_meta), _p(b_scales), _p(c),
                                                  c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()),
                                                  c_int(num_bits), c_int(size_m), c_int(size_n), c_int(size_k),
                                                  c_int(b_scales.size(0)), c_int(_dt(a)), _stream(a)))
    return c


def gptq_marlin_24_gemm_deferred(a: torch.Tensor, b_q_weight: torch.Tensor, b_meta: torch.Tensor, b_scales: torch.Tensor,
                                 workspace: torch.Tensor, num_bits: int, size_m: int, size_n: int, size_k: int) -> DeferredGemm:
    """gptq_marlin_24_gemm whose split-K partial sums are left for the consumer op, like gptq_marlin_gemm_deferred
    (argument checks: the C entry's; use the plain op to get the reference's messages)."""
    _dev(a)
    if a.dtype != torch.float16 or a.dim() != 2 or a.shape[0] != size_m or a.shape[1] != size_k or not a.is_contiguous():
        raise RuntimeError("gptq_marlin_24_gemm: a must be a contiguous fp16 [size_m, size_k] tensor")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return DeferredGemm(c, None, 1)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    splits = c_int(1)
    _lib.check(_lib.lib().nmx_gptq_marlin_24_gemm_deferred(_p(a), _p(b_q_weight), _p(b_meta), _p(b_scales), _p(c),
                                                           c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()),
                                                           c_int(num_bits), c_int(size_m), c_int(size_n), c_int(size_k),
                                                           c_int(b_scales.size(0)), c_int(_dt(a)), ctypes.byref(splits),
                                                           _stream(a)))
    if (splits.value & 0xff) > 1:
        partial, n = _slabs(scratch, splits.value, size_m, size_n)
        return DeferredGemm(c, partial, n)
    return DeferredGemm(c, None, 1)


def fp8_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor, workspace: torch.Tensor,
                    num_bits: int, size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    # checks mirror csrc/quantization/fp8/fp8_marlin.cu:1212-1280
    if num_bits != 8:
        raise RuntimeError(f"num_bits must be 8 for fp8 marlin. Got = {num_bits}")
    if a.size(0) != size_m:
        raise RuntimeError(f"Shape mismatch: a.size(0) = {a.size(0)}, size_m = {size_m}")
    if a.size(1) != size_k:
        raise RuntimeError(f"Shape mismatch: a.size(1) = {a.size(1)}, size_k = {size_k}")
    if (size_k // 16) != b_q_weight.size(0):
        raise RuntimeError(f"Shape mismatch: b_q_weight.size(0) = {b_q_weight.size(0)}, size_k = {size_k}, "
                           "tile_size = 16")
    if b_scales.dim() != 2 or b_scales.size(1) != size_n or b_scales.size(0) != 1:
        raise RuntimeError(f"b_scales must be [1, {size_n}] (channel-wise only)")
    if a.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("fp8_marlin_gemm only supports bfloat16 and float16")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    _lib.check(_lib.lib().nmx_fp8_marlin_gemm(_p(a.contiguous()), _p(b_q_weight), _p(b_scales.contiguous()), _p(c),
                                              c_i64(workspace.numel()), _p(scratch), c_i64(scratch.numel()),
                                              c_int(num_bits), c_int(size_m), c_int(size_n), c_int(size_k),
                                              c_int(_dt(a)), _stream(a)))
    return c


# ---------------------------------------------------------------------------------------------------------
# AWQ / GPTQ (vllm/_custom_ops.py:166-191)
# ---------------------------------------------------------------------------------------------------------
def _zp_scratch(a: torch.Tensor, m: int, n: int) -> torch.Tensor:
    return _get_scratch(a.device, int(_lib.lib().nmx_zp_gemm_scratch_bytes(c_int(m), c_int(n))))


_lib_zp_init = False


def _zp_lib():
    global _lib_zp_init
    lib = _lib.lib()
    if not _lib_zp_init:
        lib.nmx_zp_gemm_scratch_bytes.restype = ctypes.c_int64
        _lib_zp_init = True
    return lib


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, split_k_iters: int, thx: int,
                   thy: int) -> torch.Tensor:
    # csrc/quantization/awq/gemm_kernels.cu:436-484 (thx / thy / split_k_iters only shape the reference's launch)
    _dev(qweight)
    in_c, qout_c = qweight.shape
    if scales.dtype != torch.float16:
        raise RuntimeError("awq_dequantize: scaling factors must be float16")
    out = torch.empty((in_c, qout_c * 8), dtype=scales.dtype, device=scales.device)
    _lib.check(_lib.lib().nmx_awq_dequantize(_p(qweight), _p(scales), _p(zeros), _p(out), c_int(in_c), c_int(qout_c),
                                             c_int(in_c // scales.shape[0]), _stream(qweight)))
    return out


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
             split_k_iters: int) -> torch.Tensor:
    """Positional order is what matters: the C++ op is (in, kernel, scaling_factors, zeros, split_k)
    (csrc/ops.h:66-68) and the reference caller passes (x, qweight, scales, qzeros, pack_factor)
    (awq.py:172) into these mis-named parameters. Both orders are accepted here (told apart by dtype)."""
    _dev(input)
    scaling_factors, zeros = (qzeros, scales) if qzeros.dtype == torch.float16 else (scales, qzeros)
    if input.dtype != torch.float16 or scaling_factors.dtype != torch.float16:
        raise RuntimeError("awq_gemm only supports float16")
    m, k = input.shape
    oc = qweight.shape[1] * 8
    group_size = k // scaling_factors.shape[0]
    out = torch.empty((m, oc), dtype=input.dtype, device=input.device)
    x = input if input.is_contiguous() else input.contiguous()
    scratch = _zp_scratch(input, m, oc) if m > 0 else None
    _zp_lib()
    _lib.check(_lib.lib().nmx_awq_gemm(_p(x), _p(qweight), _p(scaling_factors), _p(zeros), _p(out), _p(scratch),
                                       c_i64(scratch.numel() if scratch is not None else 0), c_int(m), c_int(k),
                                       c_int(oc), c_int(group_size), _stream(input)))
    return out


def awq_marlin_supported(size_n: int, size_k: int, num_groups: int) -> bool:
    return bool(_lib.lib().nmx_awq_marlin_supported(c_int(size_n), c_int(size_k), c_int(num_groups)))


def awq_marlin_repack(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor):
    """One-time re-layout of an AWQ checkpoint tensor triple for awq_marlin_gemm (extension; see include/nmx.h).
    Returns (marlin_q [K / 16, 2 N] int32, marlin_scales [G, N], marlin_zeros [G, N]) fp16."""
    _dev(qweight)
    k, n = qweight.shape[0], qweight.shape[1] * 8
    groups = scales.shape[0]
    if scales.dtype != torch.float16 or scales.shape[1] != n or qzeros.shape != (groups, n // 8):
        raise RuntimeError("awq_marlin_repack: scales [G, N] fp16 and qzeros [G, N / 8] expected")
    out_q = torch.empty((k // 16, n * 2), dtype=torch.int32, device=qweight.device)
    out_s = torch.empty((groups, n), dtype=torch.float16, device=qweight.device)
    out_z = torch.empty((groups, n), dtype=torch.float16, device=qweight.device)
    _lib.check(_lib.lib().nmx_awq_marlin_repack(_p(qweight.contiguous()), _p(qzeros.contiguous()), _p(scales.contiguous()),
                                                _p(out_q), _p(out_s), _p(out_z), c_int(k), c_int(n), c_int(groups),
                                                _stream(qweight)))
    return out_q, out_s, out_z


def awq_marlin_gemm(a: torch.Tensor, marlin_q: torch.Tensor, marlin_scales: torch.Tensor, marlin_zeros: torch.Tensor,
                    size_m: int, size_n: int, size_k: int) -> torch.Tensor:
    _dev(a)
    if a.dtype != torch.float16 or a.dim() != 2 or a.shape[0] != size_m or a.shape[1] != size_k or not a.is_contiguous():
        raise RuntimeError("awq_marlin_gemm: a must be a contiguous fp16 [size_m, size_k] tensor")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return c
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    _lib.check(_lib.lib().nmx_awq_marlin_gemm(_p(a), _p(marlin_q), _p(marlin_scales), _p(marlin_zeros), _p(c), _p(scratch),
                                              c_i64(scratch.numel()), c_int(size_m), c_int(size_n), c_int(size_k),
                                              c_int(marlin_scales.shape[0]), _stream(a)))
    return c


def awq_marlin_gemm_deferred(a: torch.Tensor, marlin_q: torch.Tensor, marlin_scales: torch.Tensor, marlin_zeros: torch.Tensor,
                             size_m: int, size_n: int, size_k: int) -> DeferredGemm:
    """awq_marlin_gemm whose split-K partial sums are left for the consumer op (fused_add_rms_norm_splitk, silu_and_mul_splitk,
    rope_reshape_and_cache), like gptq_marlin_gemm_deferred."""
    _dev(a)
    if a.dtype != torch.float16 or a.dim() != 2 or a.shape[0] != size_m or a.shape[1] != size_k or not a.is_contiguous():
        raise RuntimeError("awq_marlin_gemm: a must be a contiguous fp16 [size_m, size_k] tensor")
    c = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    if size_m == 0:
        return DeferredGemm(c, None, 1)
    scratch = _marlin_scratch(a, size_m, size_n, size_k)
    splits = c_int(1)
    _lib.check(_lib.lib().nmx_awq_marlin_gemm_deferred(_p(a), _p(marlin_q), _p(marlin_scales), _p(marlin_zeros), _p(c),
                                                       _p(scratch), c_i64(scratch.numel()), c_int(size_m), c_int(size_n),
                                                       c_int(size_k), c_int(marlin_scales.shape[0]), ctypes.byref(splits),
                                                       _stream(a)))
    if (splits.value & 0xff) > 1:
        partial, n = _slabs(scratch, splits.value, size_m, size_n)
        return DeferredGemm(c, partial, n)
    return DeferredGemm(c, None, 1)


def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor, b_gptq_scales: torch.Tensor,
              b_g_idx: torch.Tensor, use_exllama: bool, bit: int) -> torch.Tensor:
    # csrc/quantization/gptq/q_gemm.cu:1823-1848
    _dev(a)
    if a.dtype != torch.float16:
        raise RuntimeError("gptq_gemm only supports float16")
    m, k = a.shape
    n = b_q_weight.shape[1]
    c = torch.empty((m, n), dtype=a.dtype, device=a.device)
    has_idx = (not b_g_idx.is_meta) and b_g_idx.numel() > 0
    if has_idx and b_g_idx.dtype != torch.int32:
        b_g_idx = b_g_idx.to(torch.int32)
    x = a if a.is_contiguous() else a.contiguous()
    scratch = _zp_scratch(a, m, n) if m > 0 else None
    _zp_lib()
    _lib.check(_lib.lib().nmx_gptq_gemm(_p(x), _p(b_q_weight), _p(b_gptq_qzeros), _p(b_gptq_scales),
                                        _p(b_g_idx if has_idx else None), _p(c), _p(scratch),
                                        c_i64(scratch.numel() if scratch is not None else 0), c_int(m), c_int(n),
                                        c_int(k), c_int(b_gptq_qzeros.shape[0]), c_int(int(bool(use_exllama))),
                                        c_int(bit), _stream(a)))
    return c


def gptq_shuffle(q_weight: torch.Tensor, q_perm: torch.Tensor, bit: int) -> None:
    # csrc/quantization/gptq/q_gemm.cu:1850-1858 (in place)
    _dev(q_weight)
    has_perm = (not q_perm.is_meta) and q_perm.numel() > 0
    if has_perm and q_perm.dtype != torch.int32:
        q_perm = q_perm.to(torch.int32)
    tmp = torch.empty_like(q_weight) if has_perm else None
    _lib.check(_lib.lib().nmx_gptq_shuffle(_p(q_weight), _p(tmp), _p(q_perm if has_perm else None),
                                           c_int(q_weight.shape[0] * 32 // bit), c_int(q_weight.shape[1]), c_int(bit),
                                           _stream(q_weight)))


# ---------------------------------------------------------------------------------------------------------
# fp8 / int8 activation quantisation and the W8A8 scaled GEMM (vllm/_custom_ops.py:218-350)
# ---------------------------------------------------------------------------------------------------------
def cutlass_scaled_mm_supports_fp8(cuda_device_capability: int) -> bool:
    return bool(_lib.lib().nmx_scaled_mm_supports_fp8(c_int(cuda_device_capability)))


def cutlass_scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                      out_dtype: Type[torch.dtype], bias: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out: written in place when given (the `Tensor! out` form of the registered op, scaled_mm_entry.cu:47-100)."""
    assert (b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0)
    assert (out_dtype is torch.bfloat16 or out_dtype is torch.float16)
    _dev(a)
    m, n, k = a.shape[0], b.shape[1], a.shape[1]
    # checks mirror csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:59-76
    if not (a.dim() == 2 and b.dim() == 2 and a.size(1) == b.size(0)):
        raise RuntimeError("cutlass_scaled_mm: a [M,K] and b [K,N] expected")
    if a.stride(1) != 1:
        raise RuntimeError("cutlass_scaled_mm: a must be row-major")
    if b.stride(0) != 1:
        raise RuntimeError("cutlass_scaled_mm: b must be column-major")
    if a.dtype != b.dtype or a.dtype not in (torch.float8_e4m3fn, torch.int8):
        raise RuntimeError("cutlass_scaled_mm: a and b must both be float8_e4m3fn or int8")
    if not (scale_a.is_contiguous() and scale_b.is_contiguous()):
        raise RuntimeError("cutlass_scaled_mm: scales must be contiguous")
    if scale_a.dtype != torch.float32 or scale_b.dtype != torch.float32:
        raise RuntimeError("cutlass_scaled_mm: scales must be float32")
    if bias is not None and not (bias.numel() == n and bias.is_contiguous() and bias.dim() == 1 and bias.dtype == out_dtype):
        raise RuntimeError("cutlass_scaled_mm: bias must be a contiguous [N] tensor of the output dtype")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif not (out.dim() == 2 and out.shape[0] == m and out.shape[1] == n and out.stride(1) == 1 and out.dtype == out_dtype):
        raise RuntimeError("cutlass_scaled_mm: out must be a row-major [M, N] tensor of the output dtype")
    lib = _lib.lib()
    scratch = _get_scratch(a.device, int(lib.nmx_scaled_mm_scratch_bytes(c_int(m), c_int(n), c_int(k))))
    _lib.check(_lib.lib().nmx_scaled_mm(_p(out), _p(a), _p(b), _p(scale_a), c_int(scale_a.numel()), _p(scale_b),
                                        c_int(scale_b.numel()), _p(bias), _p(scratch), c_i64(scratch.numel()),
                                        c_int(m), c_int(n), c_int(k),
                                        c_i64(a.stride(0)), c_i64(b.stride(1)), c_i64(out.stride(0)),
                                        c_int(int(a.dtype == torch.float8_e4m3fn)), c_int(_dt(out)), _stream(a)))
    return out


def cutlass_scaled_mm_deferred(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                               out_dtype: Type[torch.dtype]) -> DeferredGemm:
    """fp8 x fp8 cutlass_scaled_mm with per-tensor scales whose K-split reduce and scale epilogue are left to the consumer
    op (fused_add_rms_norm_splitk / silu_and_mul_splitk / rope_reshape_and_cache). The slabs live in the stream's scratch
    buffer: consume the result before the next op that uses that scratch."""
    _dev(a)
    if not (a.dtype == b.dtype == torch.float8_e4m3fn and a.dim() == 2 and b.dim() == 2 and a.size(1) == b.size(0)):
        raise RuntimeError("cutlass_scaled_mm_deferred: fp8 a [M,K] and b [K,N] expected")
    if a.stride(1) != 1 or b.stride(0) != 1:
        raise RuntimeError("cutlass_scaled_mm_deferred: a row-major, b column-major")
    if scale_a.numel() != 1 or scale_b.numel() != 1 or scale_a.dtype != torch.float32 or scale_b.dtype != torch.float32:
        raise RuntimeError("cutlass_scaled_mm_deferred: per-tensor float32 scales only")
    m, n, k = a.shape[0], b.shape[1], a.shape[1]
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    if m == 0:
        return DeferredGemm(out, None, 1)
    lib = _lib.lib()
    scratch = _get_scratch(a.device, int(lib.nmx_scaled_mm_scratch_bytes(c_int(m), c_int(n), c_int(k))))
    splits = c_int(1)
    _lib.check(lib.nmx_scaled_mm_deferred(_p(out), _p(a), _p(b), _p(scale_a), _p(scale_b), _p(scratch), c_i64(scratch.numel()),
                                          c_int(m), c_int(n), c_int(k), c_i64(a.stride(0)), c_i64(b.stride(1)),
                                          c_i64(out.stride(0)), c_int(_dt(out)), ctypes.byref(splits), _stream(a)))
    if splits.value > 1:
        partial = scratch[:splits.value * m * n * 4].view(torch.float32).view(splits.value, m, n)
        return DeferredGemm(out, partial, splits.value, scale_a, scale_b)
    return DeferredGemm(out, None, 1)


def scaled_fp8_quant(
    input: torch.Tensor,
    scale: Optional[torch.Tensor] = None,
    batch_dim_padding: Optional[int] = None,
    out: Optional[torch.Tensor] = None,
    dynamic_scale_out: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same contract as vllm/_custom_ops.py:284-320 (padding rows, if any, are left uninitialised).
    out / dynamic_scale_out: the `Tensor!` operands of static_/dynamic_scaled_fp8_quant (fp8/common.cu:129-165),
    written in place by the kernel."""
    _dev(input)
    if out is not None:
        if not (out.is_contiguous() and out.dtype == torch.float8_e4m3fn and out.numel() >= input.numel()):
            raise RuntimeError("scaled_fp8_quant: out must be a contiguous float8_e4m3fn tensor of at least input's size")
        output = out
    elif batch_dim_padding:
        shape = (max(batch_dim_padding, input.shape[0]), *input.shape[1:])
        output = torch.empty(shape, device=input.device, dtype=torch.float8_e4m3fn)
    else:
        output = torch.empty_like(input, dtype=torch.float8_e4m3fn)
    x = input if input.is_contiguous() else input.contiguous()
    if scale is None:
        # written by the kernel; no zero fill needed
        scale = dynamic_scale_out if dynamic_scale_out is not None else torch.empty(1, device=input.device, dtype=torch.float32)
        if scale.dtype != torch.float32 or scale.numel() != 1:
            raise RuntimeError("scaled_fp8_quant: scale must be a float32 scalar tensor")
        dynamic = 1
    else:
        dynamic = 0
        if scale.dtype != torch.float32 or scale.numel() != 1:
            raise RuntimeError("scaled_fp8_quant: scale must be a float32 scalar tensor")
    scratch = _get_scratch(x.device, 256)
    _lib.check(_lib.lib().nmx_scaled_fp8_quant(_p(output), _p(x), _p(scale), _p(scratch), c_i64(scratch.numel()),
                                               c_i64(x.numel()), c_int(_dt(x)), c_int(dynamic), _stream(x)))
    return output, scale


def rms_norm_absmax(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> torch.Tensor:
    """rms_norm that also returns absmax[t] = max |out[t, :]| (float32 [T]) for scaled_fp8_quant_partials."""
    _dev(input)
    hidden = input.shape[-1]
    T = input.numel() // hidden
    amax = torch.empty(T, dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().nmx_rms_norm_absmax(_p(out), _p(input), _p(weight), c_f(epsilon), c_int(T), c_int(hidden),
                                              c_int(_dt(input)), _p(amax), _stream(input)))
    return amax


def fused_add_rms_norm_absmax(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, epsilon: float) -> torch.Tensor:
    """fused_add_rms_norm (in place on input / residual) that also returns the per-token |max| of the normed output."""
    _dev(input)
    hidden = input.shape[-1]
    T = input.numel() // hidden
    amax = torch.empty(T, dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().nmx_fused_add_rms_norm_absmax(_p(input), _p(residual), _p(weight), c_f(epsilon), c_int(T),
                                                        c_int(hidden), c_int(_dt(input)), _p(amax), _stream(input)))
    return amax


def silu_and_mul_absmax(out: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """silu_and_mul that also returns the per-token |max| of its output."""
    _dev(x)
    d = x.shape[-1] // 2
    T = x.numel() // x.shape[-1]
    amax = torch.empty(T, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().nmx_act_and_mul_absmax(_p(out), _p(x), c_int(T), c_int(d), c_int(0), c_int(_dt(x)), _p(amax),
                                                 _stream(x)))
    return amax


def scaled_fp8_quant_partials(input: torch.Tensor, partials: torch.Tensor, out: Optional[torch.Tensor] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dynamic per-tensor fp8 quantisation in ONE launch, from the maxima the producer of `input` left (rms_norm_absmax,
    fused_add_rms_norm_absmax, silu_and_mul_absmax). Same (codes, scale) as scaled_fp8_quant(input), bit for bit."""
    _dev(input)
    if partials.dtype != torch.float32 or not partials.is_contiguous():
        raise RuntimeError("scaled_fp8_quant_partials: partials must be a contiguous float32 tensor")
    x = input if input.is_contiguous() else input.contiguous()
    output = out if out is not None else torch.empty_like(x, dtype=torch.float8_e4m3fn)
    scale = torch.empty(1, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().nmx_scaled_fp8_quant_partials(_p(output), _p(x), _p(scale), _p(partials), c_int(partials.numel()),
                                                        c_i64(x.numel()), c_int(_dt(x)), _stream(x)))
    return output, scale


def scaled_int8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                      dynamic_scale_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same contract as vllm/_custom_ops.py:324-350. out / dynamic_scale_out: the `Tensor!` operands of
    static_/dynamic_scaled_int8_quant (int8_quant_kernels.cu:75-115), written in place by the kernel."""
    _dev(input)
    if not input.is_contiguous():
        raise RuntimeError("input must be contiguous")  # int8_quant_kernels.cu:78
    if out is None:
        output = torch.empty_like(input, dtype=torch.int8)
    else:
        if not (out.is_contiguous() and out.dtype == torch.int8 and out.numel() == input.numel()):
            raise RuntimeError("out must be a contiguous int8 tensor of input's size")
        output = out
    hidden = input.shape[-1]
    tokens = input.numel() // hidden
    if scale is not None:
        if scale.numel() != 1 or scale.dtype != torch.float32:
            raise RuntimeError("scale.numel() == 1 (float32) expected")
        _lib.check(_lib.lib().nmx_scaled_int8_quant(_p(output), _p(input), _p(scale), c_int(tokens), c_int(hidden),
                                                    c_int(_dt(input)), c_int(0), _stream(input)))
        return output, scale
    if dynamic_scale_out is not None:
        if not (dynamic_scale_out.is_contiguous() and dynamic_scale_out.dtype == torch.float32 and dynamic_scale_out.numel() == tokens):
            raise RuntimeError("scale must be a contiguous float32 tensor with one element per token")
        input_scales = dynamic_scale_out
    else:
        input_scales = torch.empty((tokens, 1), device=input.device, dtype=torch.float32)
    _lib.check(_lib.lib().nmx_scaled_int8_quant(_p(output), _p(input), _p(input_scales), c_int(tokens), c_int(hidden),
                                                c_int(_dt(input)), c_int(1), _stream(input)))
    return output, input_scales


# ---------------------------------------------------------------------------------------------------------
# mixture-of-experts routing (vllm/_custom_ops.py:354-367)
# ---------------------------------------------------------------------------------------------------------
def moe_align_block_size(topk_ids: torch.Tensor, num_experts: int, block_size: int, sorted_token_ids: torch.Tensor,
                         experts_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor) -> None:
    _dev(topk_ids)
    if topk_ids.dtype != torch.int32 or not topk_ids.is_contiguous():
        raise RuntimeError("moe_align_block_size: topk_ids must be a contiguous int32 tensor")
    _lib.check(_lib.lib().nmx_moe_align_block_size(_p(topk_ids), c_int(num_experts), c_int(block_size), c_int(topk_ids.numel()),
                                                   _p(sorted_token_ids), c_int(sorted_token_ids.numel()), _p(experts_ids),
                                                   _p(num_tokens_post_pad), _stream(topk_ids)))


def moe_scaled_mm(out: torch.Tensor, a: torch.Tensor, w: torch.Tensor, a_scale: torch.Tensor, w_scale: torch.Tensor,
                  topk_weights: Optional[torch.Tensor], sorted_token_ids: torch.Tensor, expert_ids: torch.Tensor,
                  num_tokens_post_padded: torch.Tensor, a_row_div: int, block_rows: int) -> None:
    """Grouped fp8 GEMM over the expert-sorted (token, k) pairs (the reference's Triton fused_moe_kernel with use_fp8,
    fused_moe.py:20-292): out[id] = (a[id // a_row_div] @ w[expert].T) * [topk_weights[id]] * a_scale * w_scale[expert].
    a [rows, K] fp8, w [E, N, K] fp8, out [num_valid, N] fp16 / bf16. No host reads: graph-capturable."""
    _dev(a)
    if a.dtype != torch.float8_e4m3fn or w.dtype != torch.float8_e4m3fn or not (a.is_contiguous() and w.is_contiguous()):
        raise RuntimeError("moe_scaled_mm: a and w must be contiguous float8_e4m3fn tensors")
    if w.dim() != 3 or a.dim() != 2 or w.shape[2] != a.shape[1] or out.shape[1] != w.shape[1] or not out.is_contiguous():
        raise RuntimeError("moe_scaled_mm: shapes a [rows, K], w [E, N, K], out [num_valid, N]")
    for t in (sorted_token_ids, expert_ids, num_tokens_post_padded):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("moe_scaled_mm: index tensors must be contiguous int32")
    tw = None
    if topk_weights is not None:
        tw = topk_weights.reshape(-1)
        if tw.dtype != torch.float32 or not tw.is_contiguous() or tw.numel() != out.shape[0]:
            raise RuntimeError("moe_scaled_mm: topk_weights must be float32 with one entry per output row")
    _lib.check(_lib.lib().nmx_moe_scaled_mm(
        _p(out), _p(a), _p(w), _p(a_scale), _p(w_scale), _p(tw), _p(sorted_token_ids), _p(expert_ids),
        _p(num_tokens_post_padded), c_int(out.shape[0]), c_int(a.shape[0]), c_int(a_row_div), c_int(w.shape[1]), c_int(w.shape[2]),
        c_int(w.shape[0]), c_int(block_rows), c_int(expert_ids.numel()), c_int(_dt(out)), _stream(a)))


def moe_mm(out: torch.Tensor, a: torch.Tensor, w: torch.Tensor, topk_weights: Optional[torch.Tensor], sorted_token_ids: torch.Tensor,
           expert_ids: torch.Tensor, num_tokens_post_padded: torch.Tensor, a_row_div: int, block_rows: int) -> None:
    """Grouped fp16 / bf16 GEMM over the expert-sorted (token, k) pairs (the reference's Triton fused_moe_kernel with
    use_fp8 = False, fused_moe.py:20-292): out[id] = (a[id // a_row_div] @ w[expert].T) * [topk_weights[id]].
    a [rows, K], w [E, N, K], out [num_valid, N], all the same dtype. No host reads: graph-capturable."""
    _dev(a)
    if a.dtype not in (torch.float16, torch.bfloat16) or w.dtype != a.dtype or out.dtype != a.dtype or not (a.is_contiguous() and w.is_contiguous()):
        raise RuntimeError("moe_mm: a, w and out must be contiguous float16 / bfloat16 tensors of one dtype")
    if w.dim() != 3 or a.dim() != 2 or w.shape[2] != a.shape[1] or out.shape[1] != w.shape[1] or not out.is_contiguous():
        raise RuntimeError("moe_mm: shapes a [rows, K], w [E, N, K], out [num_valid, N]")
    for t in (sorted_token_ids, expert_ids, num_tokens_post_padded):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("moe_mm: index tensors must be contiguous int32")
    tw = None
    if topk_weights is not None:
        tw = topk_weights.reshape(-1)
        if tw.dtype != torch.float32 or not tw.is_contiguous() or tw.numel() != out.shape[0]:
            raise RuntimeError("moe_mm: topk_weights must be float32 with one entry per output row")
    _lib.check(_lib.lib().nmx_moe_mm(
        _p(out), _p(a), _p(w), _p(tw), _p(sorted_token_ids), _p(expert_ids), _p(num_tokens_post_padded), c_int(out.shape[0]),
        c_int(a.shape[0]), c_int(a_row_div), c_int(w.shape[1]), c_int(w.shape[2]), c_int(w.shape[0]), c_int(block_rows),
        c_int(expert_ids.numel()), c_int(_dt(out)), _stream(a)))


def topk_softmax(topk_weights: torch.Tensor, topk_ids: torch.Tensor, token_expert_indicies: torch.Tensor,
                 gating_output: torch.Tensor) -> None:
    _dev(gating_output)
    if gating_output.dtype != torch.float32 or not gating_output.is_contiguous() or gating_output.dim() != 2:
        raise RuntimeError("topk_softmax: gating_output must be a contiguous float32 [num_tokens, num_experts] tensor")
    t, e = gating_output.shape
    k = topk_ids.shape[-1]
    _lib.check(_lib.lib().nmx_topk_softmax(_p(topk_weights), _p(topk_ids), _p(token_expert_indicies), _p(gating_output),
                                           c_int(t), c_int(e), c_int(k), _stream(gating_output)))


# ---------------------------------------------------------------------------------------------------------
# device utilities (vllm/_custom_ops.py:415-422)
# ---------------------------------------------------------------------------------------------------------
def get_device_attribute(attribute: int, device: int) -> int:
    v = c_int(0)
    _lib.check(_lib.lib().nmx_get_device_attribute(c_int(attribute), c_int(device), ctypes.byref(v)))
    return v.value


def get_max_shared_memory_per_block_device_attribute(device: int) -> int:
    v = c_int(0)
    _lib.check(_lib.lib().nmx_get_max_shared_memory_per_block_device_attribute(c_int(device), ctypes.byref(v)))
    return v.value
