"""Shared helpers for the test-suite (fixture loading, tensor factories mirroring the reference's test helpers)."""
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DTYPES = {"float32": torch.float32, "f32": torch.float32, "float16": torch.float16, "f16": torch.float16,
          "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def from_bits(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """uint16 bit patterns -> fp16/bf16 tensor; other arrays pass through."""
    if dtype in (torch.float16, torch.bfloat16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    return torch.from_numpy(a.copy())


def seed_all(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def create_kv_caches_with_random(num_blocks: int, block_size: int, num_layers: int, num_heads: int, head_size: int,
                                 cache_dtype: str, model_dtype: torch.dtype, seed: int = 0, device: str = "cpu"
                                 ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Same shapes and value range as the reference's kv_cache_factory (vllm/utils.py:515-563):
    K [NB, H, D/x, BS, x], V [NB, H, D, BS], U(-s, s) with s = head_size^-0.5; fp8 caches are uint8."""
    seed_all(seed)
    scale = head_size**-0.5
    fp8 = cache_dtype != "auto"
    store = torch.uint8 if fp8 else model_dtype
    x = 16 // torch.tensor([], dtype=store).element_size()
    key_caches, value_caches = [], []
    for _ in range(num_layers):
        k = torch.empty(num_blocks, num_heads, head_size // x, block_size, x, dtype=torch.float32).uniform_(-scale, scale)
        v = torch.empty(num_blocks, num_heads, head_size, block_size, dtype=torch.float32).uniform_(-scale, scale)
        if fp8:
            fdt = torch.float8_e4m3fn if cache_dtype in ("fp8", "fp8_e4m3") else torch.float8_e5m2
            k = k.to(fdt).view(torch.uint8)
            v = v.to(fdt).view(torch.uint8)
        else:
            k, v = k.to(model_dtype), v.to(model_dtype)
        key_caches.append(k.to(device))
        value_caches.append(v.to(device))
    return key_caches, value_caches


def ref_single_query_cached_kv_attention(query, num_queries_per_kv, key_cache, value_cache, block_tables, seq_lens,
                                         scale, alibi_slopes: Optional[torch.Tensor]) -> torch.Tensor:
    """fp32 torch restatement of the reference test's expected value (tests/kernels/test_attention.py:63-116):
    gather keys/values through the block table, plain softmax attention."""
    num_seqs, num_heads, head_size = query.shape
    num_kv_heads = value_cache.shape[1]
    block_size = value_cache.shape[3]
    out = torch.empty(num_seqs, num_heads, head_size, dtype=torch.float32)
    kc = key_cache.float()
    vc = value_cache.float()
    for i in range(num_seqs):
        L = int(seq_lens[i])
        idx = torch.arange(L)
        blk = block_tables[i, idx // block_size].long()
        off = idx % block_size
        keys = kc[blk, :, :, off, :].reshape(L, num_kv_heads, head_size)        # [L, KVH, D/x, x] -> D
        values = vc[blk, :, :, off]                                              # [L, KVH, D]
        if num_queries_per_kv > 1:
            keys = keys.repeat_interleave(num_queries_per_kv, dim=1)
            values = values.repeat_interleave(num_queries_per_kv, dim=1)
        q = query[i].float()
        logits = scale * torch.einsum("hd,lhd->hl", q, keys)
        if alibi_slopes is not None:
            pos = torch.arange(L).float() - (L - 1)
            logits = logits + alibi_slopes.float()[:, None] * pos[None, :]
        p = torch.softmax(logits, dim=-1)
        out[i] = torch.einsum("hl,lhd->hd", p, values)
    return out


def compute_max_diff(output: torch.Tensor, output_ref: torch.Tensor) -> float:
    """mean|out - ref| / mean|ref| (reference: utils/marlin_utils.py:208-210)."""
    return float(torch.mean(torch.abs(output.float() - output_ref.float())) / torch.mean(torch.abs(output_ref.float())))


def ref_prefix_prefill(q, k, v, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, b_ctx_len, alibi_slopes=None,
                       sliding_window=0):
    """fp32 torch restatement of the expected value of the reference's test (tests/kernels/test_prefix_prefill.py:
    xformers attention with BlockDiagonalCausalFromBottomRightMask [+ local window] over context + new tokens)."""
    T, H, D = q.shape
    Hkv = k.shape[1]
    BS = v_cache.shape[3]
    out = torch.zeros(T, H, D, dtype=torch.float32)
    scale = 1.0 / (D**0.5)
    for b in range(len(b_seq_len)):
        ctx, n = int(b_ctx_len[b]), int(b_seq_len[b]) - int(b_ctx_len[b])
        s0 = int(b_start_loc[b])
        keys = torch.zeros(ctx + n, Hkv, D)
        vals = torch.zeros(ctx + n, Hkv, D)
        for j in range(ctx):
            blk, off = int(b_loc[b, j // BS]), j % BS
            keys[j] = k_cache[blk, :, :, off, :].reshape(Hkv, D).float()
            vals[j] = v_cache[blk, :, :, off].float()
        keys[ctx:] = k[s0:s0 + n].float()
        vals[ctx:] = v[s0:s0 + n].float()
        rep = H // Hkv
        kk = keys.repeat_interleave(rep, dim=1)  # [L, H, D]
        vv = vals.repeat_interleave(rep, dim=1)
        logits = torch.einsum("ihd,jhd->hij", q[s0:s0 + n].float(), kk) * scale
        qpos = torch.arange(ctx, ctx + n)[:, None]
        kpos = torch.arange(ctx + n)[None, :]
        mask = kpos > qpos
        if sliding_window and sliding_window > 0:
            mask = mask | (qpos - kpos >= sliding_window)
        if alibi_slopes is not None:
            logits = logits + alibi_slopes.float()[:, None, None] * (kpos - qpos).float()[None]
        logits = logits.masked_fill(mask[None], float("-inf"))
        out[s0:s0 + n] = torch.einsum("hij,jhd->ihd", torch.softmax(logits, dim=-1), vv)
    return out
