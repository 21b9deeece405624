"""Plain (prefix-free) prefill attention behind the two entry points ROCmFlashAttentionImpl binds for it
(vllm/attention/backends/rocm_flash_attn.py:259-282 picks `triton_attention` or CK's `flash_attn_varlen_func`; the prefill
branch :359-394 calls it with q / k / v of the prompt tokens and the cumulative sequence starts). Both run on
`nmx_context_attention_fwd` (csrc/prefill_attention.hip, the kernel behind context_attention_fwd) with an EMPTY paged context -
the kernel takes the new tokens' K / V straight from the `k` / `v` operands, causal inside every sequence, GQA / MQA without
repeat_kv. No Triton, no CK, no SDPA fall-back: a deployer sets

    vllm.attention.ops.triton_flash_attention.triton_attention = triton_attention        (VLLM_USE_TRITON_FLASH_ATTN=1, default)
    flash_attn.flash_attn_varlen_func = flash_attn_varlen_func                          (VLLM_USE_TRITON_FLASH_ATTN=0)

(INTEGRATION.md). What the reference passes and this path does not serve raises instead of silently computing something else:
non-causal calls, different q / k sequence starts (chunked prefill over a prefix goes through context_attention_fwd), a dense
additive bias (ALiBi is taken as `alibi_slopes`, which the kernel applies itself - the reference materialises a [H, L, L]
bias tensor for the Triton kernel, rocm_flash_attn.py:363-368)."""
from typing import Optional, Tuple

import torch

from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd

_DUMMY = {}


def _dummy_cache(device, dtype, kv_heads: int, head: int):
    """One never-read block in the reference's cache layout (the kernel wants the cache geometry even for an empty context)."""
    key = (device, dtype, kv_heads, head)
    if key not in _DUMMY:
        x = 16 // torch.empty(0, dtype=dtype).element_size()
        _DUMMY[key] = (torch.zeros(1, kv_heads, head // x, 16, x, dtype=dtype, device=device),
                       torch.zeros(1, kv_heads, head, 16, dtype=dtype, device=device))
    return _DUMMY[key]


def _prefill(q, k, v, o, cu_seqlens, max_seqlen: int, sm_scale: float, alibi_slopes, sliding_window):
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise RuntimeError("prefill attention: q [T, H, D], k / v [T, Hkv, D] expected")
    head = q.shape[-1]
    if abs(sm_scale * head**0.5 - 1.0) > 1e-6:
        raise RuntimeError("prefill attention: only the default softmax scale 1 / sqrt(head_size) is served")
    if o is None:
        o = torch.empty_like(q)
    cu = cu_seqlens.to(torch.int32)
    nseq = cu.numel() - 1
    k_cache, v_cache = _dummy_cache(q.device, q.dtype, k.shape[1], head)
    b_loc = torch.zeros(nseq, 1, dtype=torch.int32, device=q.device)
    b_seq_len = cu[1:] - cu[:-1]
    b_ctx_len = torch.zeros(nseq, dtype=torch.int32, device=q.device)
    context_attention_fwd(q, k, v, o, k_cache, v_cache, b_loc, cu[:-1].contiguous(), b_seq_len.contiguous(), b_ctx_len,
                          int(max_seqlen), alibi_slopes, sliding_window)
    return o


def triton_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: Optional[torch.Tensor], cu_seqlens_q: torch.Tensor,
                     cu_seqlens_k: torch.Tensor, max_seqlens_q: int, max_seqlens_k: int, causal: bool = False,
                     sm_scale: float = 1.0, bias: Optional[torch.Tensor] = None, *,
                     alibi_slopes: Optional[torch.Tensor] = None,
                     sliding_window: Optional[int] = None) -> Tuple[torch.Tensor, None]:
    """Signature of vllm/attention/ops/triton_flash_attention.py `triton_attention` (the autograd Function's apply); returns
    (out, None) like it."""
    if not causal:
        raise RuntimeError("triton_attention: the decoder prefill path is causal (rocm_flash_attn.py:377)")
    if bias is not None:
        raise RuntimeError("triton_attention: pass ALiBi as alibi_slopes= (the kernel applies the slopes itself); dense biases are not served")
    if cu_seqlens_q.data_ptr() != cu_seqlens_k.data_ptr() and not torch.equal(cu_seqlens_q, cu_seqlens_k):
        raise RuntimeError("triton_attention: q and k must be the same prompt tokens (prefix prefill: context_attention_fwd)")
    return _prefill(q, k, v, o, cu_seqlens_q, max(int(max_seqlens_q), int(max_seqlens_k)), sm_scale, alibi_slopes, sliding_window), None


def flash_attn_varlen_func(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                           max_seqlen_q: int, max_seqlen_k: int, dropout_p: float = 0.0, softmax_scale: Optional[float] = None,
                           causal: bool = False, window_size=(-1, -1), alibi_slopes: Optional[torch.Tensor] = None, **unused) -> torch.Tensor:
    """Signature of flash_attn.flash_attn_varlen_func as rocm_flash_attn.py:395-406 calls it (CK path)."""
    if dropout_p != 0.0 or not causal:
        raise RuntimeError("flash_attn_varlen_func: causal, dropout-free prefill only")
    scale = q.shape[-1]**-0.5 if softmax_scale is None else softmax_scale
    window = None if window_size is None or window_size[0] is None or window_size[0] < 0 else int(window_size[0]) + 1
    out, _ = triton_attention(q, k, v, None, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, True, scale, None,
                              alibi_slopes=alibi_slopes, sliding_window=window)
    return out
