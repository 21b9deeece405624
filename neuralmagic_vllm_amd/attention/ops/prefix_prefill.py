"""Prefill attention over paged context + new tokens — same entry point as vllm/attention/ops/prefix_prefill.py
(`context_attention_fwd`, :674-812), HIP kernel instead of Triton (csrc/prefill_attention.hip)."""
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch

from neuralmagic_vllm_amd import _lib


def _p(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


@torch.inference_mode()
def context_attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, k_cache: torch.Tensor,
                          v_cache: torch.Tensor, b_loc: torch.Tensor, b_start_loc: torch.Tensor, b_seq_len: torch.Tensor,
                          b_ctx_len: torch.Tensor, max_input_len: int, alibi_slopes: Optional[torch.Tensor] = None,
                          sliding_window: Optional[int] = None) -> None:
    if not q.is_cuda:
        raise RuntimeError("context_attention_fwd: tensors must be on the GPU")
    Lq, Lk, Lv = q.shape[-1], k.shape[-1], v.shape[-1]
    assert Lq == Lk and Lk == Lv
    if q.dtype not in (torch.float16, torch.bfloat16) or k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise RuntimeError("context_attention_fwd: float16 / bfloat16 query with a cache of the same dtype")
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.stride(-1) != 1:
            raise RuntimeError(f"context_attention_fwd: {name} must be contiguous in its last dimension")
    if k_cache.dim() != 5 or v_cache.dim() != 4 or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise RuntimeError("context_attention_fwd: k_cache [NB, Hkv, D/x, BS, x] and v_cache [NB, Hkv, D, BS] expected")
    sliding_window = 0 if sliding_window is None or sliding_window <= 0 else int(sliding_window)
    b_loc_, st_, sl_, cl_ = _i32(b_loc), _i32(b_start_loc), _i32(b_seq_len), _i32(b_ctx_len)
    al = None
    if alibi_slopes is not None:
        al = alibi_slopes.to(torch.float32).contiguous()
    dt = 1 if q.dtype == torch.float16 else 2
    _lib.check(_lib.lib().nmx_context_attention_fwd(
        _p(o), _p(q), _p(k), _p(v), _p(k_cache), _p(v_cache), _p(b_loc_), _p(st_), _p(sl_), _p(cl_), _p(al),
        c_int(sl_.shape[0]), c_int(q.shape[1]), c_int(k.shape[1]), c_int(Lq), c_int(v_cache.shape[3]),
        c_int(k_cache.shape[4]), c_int64(q.stride(0)), c_int64(q.stride(1)), c_int64(k.stride(0)), c_int64(k.stride(1)),
        c_int64(v.stride(0)), c_int64(v.stride(1)), c_int64(o.stride(0)), c_int64(o.stride(1)),
        c_int64(k_cache.stride(0)), c_int64(k_cache.stride(1)), c_int64(v_cache.stride(0)), c_int64(v_cache.stride(1)),
        c_int64(b_loc_.stride(0)), c_int(int(max_input_len)), c_int(sliding_window), c_float(1.0 / (Lq**0.5)), c_int(dt),
        c_void_p(torch.cuda.current_stream(q.device).cuda_stream)))
