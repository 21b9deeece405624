"""Caller-side shim of the paged KV cache — mirror of vllm/attention/ops/paged_attn.py:32-239 (PagedAttention):
cache shape / split, write_to_paged_cache, forward_decode (v1 / v2 choice and v2 temporaries), swap / copy."""
from typing import List, Optional, Tuple

import torch

from neuralmagic_vllm_amd import _custom_ops as ops

_PARTITION_SIZE = 512  # paged_attn.py:14


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int, head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                             slot_mapping: torch.Tensor, kv_cache_dtype: str, kv_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(), kv_cache_dtype, kv_scale)

    @staticmethod
    def forward_decode(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor, block_tables: torch.Tensor,
                       seq_lens: torch.Tensor, max_seq_len: int, kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], kv_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64, blocksparse_head_sliding_step: int = 0) -> torch.Tensor:
        if blocksparse_vert_stride is not None and blocksparse_vert_stride > 1:
            block_size = value_cache.size(-1)
            assert (blocksparse_block_size > 0 and blocksparse_block_size % block_size == 0), \
                (f"{blocksparse_block_size=} needs to be a multiple of {block_size=} used in block_tables.")
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        # paged_attn.py:112-121: v1 for short contexts or when (heads x seqs) alone fills the GPU
        use_v1 = (max_seq_len <= 8192 and (max_num_partitions == 1 or num_seqs * num_heads > 512))
        if use_v1:
            ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
                                   blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step)
        else:
            assert _PARTITION_SIZE % block_size == 0
            tmp_output = torch.empty(size=(num_seqs, num_heads, max_num_partitions, head_size), dtype=output.dtype,
                                     device=output.device)
            exp_sums = torch.empty(size=(num_seqs, num_heads, max_num_partitions), dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache, value_cache, num_kv_heads,
                                   scale, block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                                   kv_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step)
        return output

    @staticmethod
    def forward_prefix(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, block_tables: torch.Tensor, query_start_loc: torch.Tensor,
                       seq_lens_tensor: torch.Tensor, context_lens: torch.Tensor, max_query_len: int,
                       alibi_slopes: Optional[torch.Tensor], sliding_window: Optional[int]) -> torch.Tensor:
        """paged_attn.py:183-216 — prefill with a cached prefix (query_start_loc is [batch + 1])."""
        from neuralmagic_vllm_amd.attention.ops.prefix_prefill import context_attention_fwd
        output = torch.empty_like(query)
        context_attention_fwd(query, key, value, output, key_cache, value_cache, block_tables, query_start_loc[:-1],
                              seq_lens_tensor, context_lens, max_query_len, alibi_slopes, sliding_window)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor, src_to_dst: torch.Tensor) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        ops.copy_blocks([kv[0] for kv in kv_caches], [kv[1] for kv in kv_caches], src_to_dists)
