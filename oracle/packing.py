"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's weight quantise/pack utilities.

These are the *forward* direction (dense -> quantised -> packed checkpoint / kernel formats); the C++
oracle holds the inverse direction. Each function cites the reference code it restates (paths relative
to the reference root). Pinned by tests/golden/ fixtures generated from the reference's own Python
utilities (tests/golden/gen_golden.py).
"""
from typing import Optional, Tuple

import numpy as np
import torch


# ---- symmetric fake-quant: vllm/model_executor/layers/quantization/utils/quant_utils.py:39-106 ----
def quantize_weights(w: torch.Tensor, num_bits: int, group_size: int, act_order: bool = False,
                     generator: Optional[torch.Generator] = None):
    """Returns (w_ref, q_w, s, g_idx, rand_perm). s = 2*max|w|/(2^b-1) per (group, column);
    q = clamp(round(w/s) + 2^(b-1), 0, 2^b-1); w_ref = (q - 2^(b-1)).half() * s."""
    size_k, size_n = w.shape
    if group_size == -1:
        group_size = size_k
    max_q = 2**num_bits - 1
    half_q = (max_q + 1) // 2
    wg = w.reshape(size_k // group_size, group_size, size_n)
    s = wg.abs().amax(dim=1, keepdim=True)  # [G,1,N]
    s = s * (2 / max_q)
    q = torch.round(wg / s).int() + half_q
    q = torch.clamp(q, 0, max_q)
    w_ref = ((q - half_q).half() * s).reshape(size_k, size_n)
    q = q.reshape(size_k, size_n)
    s = s.reshape(-1, size_n).contiguous()
    g_idx = torch.empty(0, dtype=torch.int32)
    rand_perm = torch.empty(0, dtype=torch.int64)
    if act_order:
        assert group_size < size_k
        g_idx = (torch.arange(size_k, dtype=torch.int32) // group_size)
        rand_perm = torch.randperm(size_k, generator=generator)
        g_idx = g_idx[rand_perm].contiguous()
        q = q[rand_perm, :].contiguous()
        w_ref = w_ref[rand_perm, :].contiguous()
    return w_ref, q, s, g_idx, rand_perm


# quant_utils.py:109-122
def sort_weights(q_w: torch.Tensor, g_idx: torch.Tensor):
    sort_indices = torch.argsort(g_idx).to(torch.int32)
    return q_w[sort_indices.long(), :].contiguous(), g_idx[sort_indices.long()].contiguous(), sort_indices


# quant_utils.py:125-146: element k sits at bits (k % pf) * bits of row k // pf
def gptq_pack(q_w: torch.Tensor, num_bits: int, size_k: int, size_n: int) -> torch.Tensor:
    pf = 32 // num_bits
    q = q_w.cpu().numpy().astype(np.uint32).reshape(size_k // pf, pf, size_n)
    shifts = (np.arange(pf, dtype=np.uint32) * num_bits)[None, :, None]
    packed = np.bitwise_or.reduce(q << shifts, axis=1)
    return torch.from_numpy(np.ascontiguousarray(packed.astype(np.int32)))


def gptq_pack_zeros(z: torch.Tensor, num_bits: int) -> torch.Tensor:
    """qzeros [groups, N/pf] packed along N; stored value is z - 1 (gptq.py:141-196, q_gemm.cu:1408)."""
    pf = 32 // num_bits
    G, N = z.shape
    zz = ((z.cpu().numpy().astype(np.int64) - 1) & (2**num_bits - 1)).astype(np.uint32).reshape(G, N // pf, pf)
    shifts = (np.arange(pf, dtype=np.uint32) * num_bits)[None, None, :]
    return torch.from_numpy(np.ascontiguousarray(np.bitwise_or.reduce(zz << shifts, axis=2).astype(np.int32)))


# ---- Marlin layout: marlin_perms.py:16-50, marlin_utils.py:25-57 (element map: SURVEY.md appendix A.2) ----
def _marlin_perm(num_bits: int) -> np.ndarray:
    i = np.arange(32)
    col = i // 4
    rows = np.stack([2 * (i % 4), 2 * (i % 4) + 1, 2 * (i % 4 + 4), 2 * (i % 4 + 4) + 1], axis=1)  # [32,4]
    perm1 = np.concatenate([16 * rows + col[:, None], 16 * rows + col[:, None] + 8], axis=1)  # [32,8]
    perm = (perm1[:, None, :] + 256 * np.arange(4)[None, :, None]).reshape(-1)  # [32,4,8] -> 1024
    inter = np.array([0, 2, 4, 6, 1, 3, 5, 7]) if num_bits == 4 else np.array([0, 2, 1, 3])
    return perm.reshape(-1, len(inter))[:, inter].reshape(-1)


def marlin_weights(q_w: torch.Tensor, size_k: int, size_n: int, num_bits: int) -> torch.Tensor:
    pf = 32 // num_bits
    q = q_w.cpu().numpy().astype(np.uint32)
    q = q.reshape(size_k // 16, 16, size_n // 16, 16).transpose(0, 2, 1, 3).reshape(size_k // 16, size_n * 16)
    perm = _marlin_perm(num_bits)
    q = q.reshape(-1, 1024)[:, perm].reshape(size_k // 16, size_n * 16)
    q = q.reshape(size_k // 16, size_n * 16 // pf, pf)
    shifts = (np.arange(pf, dtype=np.uint32) * num_bits)[None, None, :]
    return torch.from_numpy(np.ascontiguousarray(np.bitwise_or.reduce(q << shifts, axis=2).astype(np.int32)))


_SCALE_PERM = [i + 8 * j for i in range(8) for j in range(8)]
_SCALE_PERM_SINGLE = [2 * i + j for i in range(4) for j in [0, 1, 8, 9, 16, 17, 24, 25]]


# marlin_utils.py:60-69 / gptq_marlin.py:47-56
def marlin_permute_scales(s: torch.Tensor, size_k: int, size_n: int, group_size: int) -> torch.Tensor:
    if group_size < size_k and group_size != -1:
        s = s.reshape(-1, 64)[:, _SCALE_PERM]
    else:
        s = s.reshape(-1, 32)[:, _SCALE_PERM_SINGLE]
    return s.reshape(-1, size_n).contiguous()


# marlin_utils.py:72-109
def marlin_quantize(w: torch.Tensor, num_bits: int, group_size: int, act_order: bool,
                    generator: Optional[torch.Generator] = None):
    size_k, size_n = w.shape
    if group_size == -1:
        group_size = size_k
    w_ref, q_w, s, g_idx, rand_perm = quantize_weights(w, num_bits, group_size, act_order, generator)
    sort_indices = torch.empty(0, dtype=torch.int32)
    if act_order:
        q_w, g_idx, sort_indices = sort_weights(q_w, g_idx)
    marlin_q_w = marlin_weights(q_w, size_k, size_n, num_bits)
    marlin_s = marlin_permute_scales(s, size_k, size_n, group_size)
    return w_ref, marlin_q_w, marlin_s, g_idx, sort_indices, rand_perm


# marlin_utils.py:226-247
def pack_fp8_to_int32(fp8_tensor: torch.Tensor) -> torch.Tensor:
    assert fp8_tensor.dtype == torch.float8_e4m3fn and fp8_tensor.shape[0] % 4 == 0
    b = fp8_tensor.view(torch.uint8).cpu().numpy().astype(np.uint32)
    b = b.reshape(b.shape[0] // 4, 4, *b.shape[1:])
    packed = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24)
    return torch.from_numpy(packed.astype(np.int32)).contiguous()


# ---- AWQ: awq.py:104-152, awq/dequantize.cuh:17-98 — element j of a group of 8 columns sits in nibble order[j] ----
_AWQ_ORDER = np.array([0, 4, 1, 5, 2, 6, 3, 7])


def awq_pack(q: torch.Tensor) -> torch.Tensor:
    """q [R, N] int in [0,15] -> [R, N/8] int32 in AWQ nibble order."""
    R, N = q.shape
    v = q.cpu().numpy().astype(np.uint32).reshape(R, N // 8, 8)
    shifts = (4 * _AWQ_ORDER).astype(np.uint32)[None, None, :]
    return torch.from_numpy(np.ascontiguousarray(np.bitwise_or.reduce(v << shifts, axis=2).astype(np.int32)))


def awq_quantize(w: torch.Tensor, group_size: int, generator: Optional[torch.Generator] = None):
    """Asymmetric 4-bit: returns (w_ref fp16 [K,N], qweight [K,N/8], qzeros [K/g,N/8], scales [K/g,N])."""
    K, N = w.shape
    wg = w.float().reshape(K // group_size, group_size, N)
    mx, mn = wg.amax(dim=1, keepdim=True), wg.amin(dim=1, keepdim=True)
    s = ((mx - mn).clamp(min=1e-5) / 15).half()
    z = torch.clamp(torch.round(-mn / s.float()), 0, 15).int()
    q = torch.clamp(torch.round(wg / s.float()) + z, 0, 15).int()
    w_ref = ((q - z).half() * s).reshape(K, N)
    return (w_ref, awq_pack(q.reshape(K, N)), awq_pack(z.reshape(K // group_size, N)),
            s.reshape(K // group_size, N).contiguous())


def gptq_quantize(w: torch.Tensor, num_bits: int, group_size: int):
    """Asymmetric GPTQ checkpoint format: (w_ref, qweight [K/pf,N], qzeros [K/g,N/pf], scales [K/g,N], g_idx [K])."""
    K, N = w.shape
    maxq = 2**num_bits - 1
    wg = w.float().reshape(K // group_size, group_size, N)
    mx, mn = wg.amax(dim=1, keepdim=True), wg.amin(dim=1, keepdim=True)
    s = ((mx - mn).clamp(min=1e-5) / maxq).half()
    z = torch.clamp(torch.round(-mn / s.float()), 1, maxq).int()  # stored as z-1 >= 0
    q = torch.clamp(torch.round(wg / s.float()) + z, 0, maxq).int()
    w_ref = ((q - z).half() * s).reshape(K, N)
    g_idx = (torch.arange(K, dtype=torch.int32) // group_size)
    return (w_ref, gptq_pack(q.reshape(K, N), num_bits, K, N), gptq_pack_zeros(z.reshape(-1, N), num_bits),
            s.reshape(-1, N).contiguous(), g_idx)
