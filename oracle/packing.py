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


def _pack_stream(v: np.ndarray, num_bits: int, axis: int) -> np.ndarray:
    """Packs integer codes along `axis` into a contiguous little-endian bit stream of num_bits-wide fields cut into
    int32 words (2 / 4 / 8 bit: 32 // bits fields per word; 3 bit: 32 fields per 3 words)."""
    v = np.moveaxis(v.astype(np.uint64), axis, 0)
    n = v.shape[0]
    assert (n * num_bits) % 32 == 0
    words = np.zeros((n * num_bits // 32, ) + v.shape[1:], dtype=np.uint64)
    for i in range(n):
        pos = i * num_bits
        w, sh = pos // 32, pos % 32
        words[w] |= (v[i] << sh) & 0xffffffff
        if sh + num_bits > 32:
            words[w + 1] |= v[i] >> (32 - sh)
    return np.moveaxis(words.astype(np.uint32).view(np.int32), 0, axis)


# quant_utils.py:125-146: element k sits at bits (k % pf) * bits of row k // pf (3 bit: AutoGPTQ's 32-in-3-words stream)
def gptq_pack(q_w: torch.Tensor, num_bits: int, size_k: int, size_n: int) -> torch.Tensor:
    q = q_w.cpu().numpy().reshape(size_k, size_n)
    return torch.from_numpy(np.ascontiguousarray(_pack_stream(q, num_bits, 0)))


def gptq_pack_zeros(z: torch.Tensor, num_bits: int) -> torch.Tensor:
    """qzeros [groups, N * bits / 32] packed along N; stored value is z - 1 (gptq.py:141-196, q_gemm.cu:1408)."""
    zz = (z.cpu().numpy().astype(np.int64) - 1) & (2**num_bits - 1)
    return torch.from_numpy(np.ascontiguousarray(_pack_stream(zz, num_bits, 1)))


# ---- Marlin layout: marlin_perms.py:16-50, marlin_utils.py:25-57 (element map: SURVEY.md appendix A.2) ----
def _marlin_perm(num_bits: int) -> np.ndarray:
    i = np.arange(32)
    col = i // 4
    rows = np.stack([2 * (i % 4), 2 * (i % 4) + 1, 2 * (i % 4 + 4), 2 * (i % 4 + 4) + 1], axis=1)  # [32,4]
    perm1 = np.concatenate([16 * rows + col[:, None], 16 * rows + col[:, None] + 8], axis=1)  # [32,8]
    perm = (perm1[:, None, :] + 256 * np.arange(4)[None, :, None]).reshape(-1)  # [32,4,8] -> 1024
    inter = np.array([0, 2, 4, 6, 1, 3, 5, 7]) if num_bits == 4 else np.array([0, 2, 1, 3])
    return perm.reshape(-1, len(inter))[:, inter].reshape(-1)


def marlin_weights(q_w: torch.Tensor, size_k: int, size_n: int, num_bits: int,
                   perm: Optional[np.ndarray] = None) -> torch.Tensor:
    pf = 32 // num_bits
    q = q_w.cpu().numpy().astype(np.uint32)
    q = q.reshape(size_k // 16, 16, size_n // 16, 16).transpose(0, 2, 1, 3).reshape(size_k // 16, size_n * 16)
    if perm is None:
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


# ---- 2:4 sparse Marlin: marlin_24_perms.py:16-50, format_24.py:21-177, marlin_utils.py:145-198 ----
def _marlin_24_perm(num_bits: int) -> np.ndarray:
    out = []
    for i in range(32):
        col, m = i // 4, i % 4
        perm1 = [16 * row + (col // 2) * 256 + 8 * (col % 2) + 4 * block
                 for block in (0, 1) for row in (2 * m, 2 * m + 1, 2 * (m + 4), 2 * (m + 4) + 1)]
        for j in range(4):
            out.extend(p + j for p in perm1)
    perm = np.array(out)
    inter = np.array([0, 2, 4, 6, 1, 3, 5, 7]) if num_bits == 4 else np.array([0, 2, 1, 3])
    return perm.reshape(-1, len(inter))[:, inter].reshape(-1)


_SCALE_PERM_24 = [i * 8 + j for i in range(8) for j in [0, 4, 1, 5, 2, 6, 3, 7]]


def marlin_24_permute_scales(s: torch.Tensor, size_k: int, size_n: int, group_size: int) -> torch.Tensor:
    if group_size < size_k and group_size != -1:
        s = s.reshape(-1, 64)[:, _SCALE_PERM_24]
    # channel-wise: marlin_24_scale_perm_single is the identity
    return s.reshape(-1, size_n).contiguous()


# format_24.py:279-308 — keep the 2 largest |w| of every 4 consecutive elements (ties: argsort order)
def mask_creator(t: torch.Tensor) -> torch.Tensor:
    tmp = t.detach().abs().reshape(-1, 4)
    index = torch.argsort(tmp, dim=1)[:, :2]
    return torch.ones_like(tmp, dtype=torch.float32).scatter_(1, index, 0).reshape(t.shape)


def _meta_reorder_offsets(m: int, meta_ncols: int) -> np.ndarray:
    """format_24.py:21-50 for int16 meta: flat destination of meta element (row, col) in the CUTLASS
    ColumnMajorInterleaved<2> reordered tensor."""
    r = np.arange(m)[:, None].repeat(meta_ncols, 1)
    c = np.arange(meta_ncols)[None, :].repeat(m, 0)
    gx, gy = 64, 32
    r = r // gx * gx + (r % 2) * 2 + (r % 8) // 4 + ((r % gy) % 4) // 2 * 32 + ((r % gx) // 8) * 4
    tr = ((r % 2 == 0) & (c % 2 == 1)).astype(np.int64)
    bl = ((r % 2 == 1) & (c % 2 == 0)).astype(np.int64)
    r = r + tr - bl
    c = c - (tr - bl)
    return ((c // 2) * m * 2 + r * 2 + c % 2).reshape(-1)


def compress_quantized_24_weight(q_24: torch.Tensor, size_k: int, size_n: int, num_bits: int):
    """marlin_utils.py:145-166 + format_24.py:56-177. q_24 [K,N] ints with zero point 2^(b-1) on pruned slots.
    Returns (q_comp [K/2,N] int32, meta [K/32, 2N] int16)."""
    zp = 1 << (num_bits - 1)
    d = (q_24.cpu().numpy().astype(np.int64) - zp).T  # [N, K]
    n, k = d.shape
    assert n % 32 == 0 and k % 16 == 0
    d4 = d.reshape(n, k // 4, 4)
    m0, m1, _, m3 = [(d4[..., i] != 0) for i in range(4)]
    e0, e1, e2 = m0 & m1, ~m0 & m1, ~m0 & ~m1
    idx0 = e1.astype(np.int64) | (e2.astype(np.int64) << 1)
    idx1 = (e0 | e2 | m3).astype(np.int64) | ((e1 | ~m1).astype(np.int64) << 1)
    s0 = np.take_along_axis(d4, idx0[..., None], -1)
    s1 = np.take_along_axis(d4, idx1[..., None], -1)
    comp = np.concatenate([s0, s1], -1).reshape(n, k // 2)
    meta4 = (idx0 | (idx1 << 2)).reshape(n, k // 16, 4)
    meta = (meta4[..., 0] | (meta4[..., 1] << 4) | (meta4[..., 2] << 8) | (meta4[..., 3] << 12)).astype(np.uint16)
    ncols = k // 16
    out = np.empty(n * ncols, dtype=np.uint16)
    out[_meta_reorder_offsets(n, ncols)] = meta.reshape(-1)
    meta_r = out.view(np.int16).reshape(ncols // 2, n * 2)  # resize_ without moving data (marlin_utils.py:162)
    q_comp = torch.from_numpy(np.ascontiguousarray(comp.T + zp).astype(np.int32))
    return q_comp, torch.from_numpy(meta_r.copy())


def marlin_24_quantize(w: torch.Tensor, num_bits: int, group_size: int):
    """marlin_utils.py:169-205 without the .cuda() hop. Returns (w_24_ref, marlin_24_q_w_comp, meta, marlin_24_s)."""
    size_k, size_n = w.shape
    if group_size == -1:
        group_size = size_k
    mask = mask_creator(w.t()).t().bool()
    w_24 = (mask * w).contiguous()
    w_24_ref, q_w_24, s, _, _ = quantize_weights(w_24, num_bits, group_size, False)
    q_comp, meta = compress_quantized_24_weight(q_w_24, size_k, size_n, num_bits)
    mq = marlin_weights(q_comp, size_k // 2, size_n, num_bits, _marlin_24_perm(num_bits))
    ms = marlin_24_permute_scales(s, size_k, size_n, group_size)
    return w_24_ref, mq, meta, ms


def marlin_24_decode(mq: torch.Tensor, meta: torch.Tensor, ms: torch.Tensor, num_bits: int, size_k: int, size_n: int,
                     group_size: int) -> torch.Tensor:
    """Inverse of marlin_24_quantize's packing: kernel-format tensors -> dense fp16 W [K,N] (what
    gptq_marlin_24_gemm multiplies by; marlin_24_cuda_kernel.cu:111-860 read as a data format)."""
    pf = 32 // num_bits
    zp = 1 << (num_bits - 1)
    kc = size_k // 2
    v = mq.cpu().numpy().astype(np.uint32)
    shifts = (np.arange(pf, dtype=np.uint32) * num_bits)[None, None, :]
    q = ((v[:, :, None] >> shifts) & (2**num_bits - 1)).reshape(kc // 16, size_n * 16).astype(np.int64)
    perm = _marlin_24_perm(num_bits)
    tiles = np.empty_like(q).reshape(-1, 1024)
    tiles[:, perm] = q.reshape(-1, 1024)
    comp = tiles.reshape(kc // 16, size_n // 16, 16, 16).transpose(0, 2, 1, 3).reshape(kc, size_n)  # [K/2, N]
    comp = comp.T - zp  # [N, K/2]
    ncols = size_k // 16
    flat = meta.cpu().numpy().reshape(-1).view(np.uint16).astype(np.int64)
    m = flat[_meta_reorder_offsets(size_n, ncols)].reshape(size_n, ncols)
    nib = np.stack([(m >> (4 * i)) & 15 for i in range(4)], -1).reshape(size_n, size_k // 4)
    dense = np.zeros((size_n, size_k // 4, 4), dtype=np.int64)
    c2 = comp.reshape(size_n, size_k // 4, 2)
    np.put_along_axis(dense, (nib & 3)[..., None], c2[..., 0:1], -1)
    # second value last so that (degenerate) equal indices resolve like the hardware: later element wins is
    # irrelevant for valid 2:4 metadata (idx0 < idx1 always)
    np.put_along_axis(dense, (nib >> 2)[..., None], c2[..., 1:2], -1)
    dq = torch.from_numpy(dense.reshape(size_n, size_k).T.copy()).half()  # [K,N] q - zp
    if group_size == -1:
        group_size = size_k
    if group_size < size_k:
        inv = np.argsort(np.array(_SCALE_PERM_24))
        s = ms.reshape(-1, 64)[:, inv].reshape(-1, size_n)
    else:
        s = ms.reshape(-1, size_n)
    return (dq.reshape(size_k // group_size, group_size, size_n) * s[:, None, :]).reshape(size_k, size_n)


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
