"""GPU parity tests for the KV-cache ops (bit-exact vs the CPU oracle and the reference-made golden vectors).
Mirrors tests/kernels/test_cache.py of the reference."""
import random

import numpy as np
import pytest
import torch

import oracle
from util import DTYPES, create_kv_caches_with_random, from_bits, load_golden, seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _int_view(t):
    return t.view({1: torch.uint8, 2: torch.int16, 4: torch.int32}[t.element_size()])


@pytest.mark.parametrize("tag", ["f32", "bf16", "bf16_opt125m"])  # opt125m: BASELINE configs[0] geometry (12 x 64 heads)
def test_reshape_and_cache_golden(ops, tag):
    g = load_golden("reshape_and_cache_" + tag)
    dt = DTYPES[tag.split("_")[0]]
    qkv = from_bits(g["qkv"], dt).to(DEV)
    kc = from_bits(g["k_cache_in"], dt).to(DEV)
    vc = from_bits(g["v_cache_in"], dt).to(DEV)
    ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, torch.from_numpy(g["slot_mapping"]).to(DEV), "auto", 1.0)
    assert torch.equal(_int_view(kc.cpu()), _int_view(from_bits(g["k_cache_out"], dt)))
    assert torch.equal(_int_view(vc.cpu()), _int_view(from_bits(g["v_cache_out"], dt)))


@pytest.mark.parametrize("num_tokens", [1, 42])
@pytest.mark.parametrize("num_heads,head_size", [(8, 64), (8, 80), (2, 128), (3, 256)])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8", "fp8_e5m2"])
def test_reshape_and_cache(ops, num_tokens, num_heads, head_size, block_size, dtype, kv_cache_dtype):
    seed_all(0)
    num_blocks = 128
    slot_mapping = torch.tensor(random.sample(range(num_blocks * block_size), num_tokens), dtype=torch.long)
    if num_tokens > 4:
        slot_mapping[3] = -1
    qkv = torch.randn(num_tokens, 3, num_heads, head_size, dtype=dtype)
    _, key, value = qkv.unbind(dim=1)
    kcs, vcs = create_kv_caches_with_random(num_blocks, block_size, 1, num_heads, head_size, kv_cache_dtype, dtype)
    kc, vc = kcs[0], vcs[0]
    kv_scale = 1.0 if kv_cache_dtype == "auto" else 0.37
    kc_o, vc_o = kc.clone(), vc.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slot_mapping, kv_cache_dtype, kv_scale)
    kc_g, vc_g = kc.to(DEV), vc.to(DEV)
    qkv_g = qkv.to(DEV)
    ops.reshape_and_cache(qkv_g[:, 1], qkv_g[:, 2], kc_g, vc_g, slot_mapping.to(DEV), kv_cache_dtype, kv_scale)
    assert torch.equal(_int_view(kc_g.cpu()), _int_view(kc_o))
    assert torch.equal(_int_view(vc_g.cpu()), _int_view(vc_o))


def test_reshape_and_cache_empty(ops):
    kc = torch.zeros(4, 2, 8, 16, 8, dtype=torch.half, device=DEV)
    vc = torch.zeros(4, 2, 64, 16, dtype=torch.half, device=DEV)
    k = torch.zeros(0, 2, 64, dtype=torch.half, device=DEV)
    ops.reshape_and_cache(k, k, kc, vc, torch.zeros(0, dtype=torch.long, device=DEV), "auto", 1.0)
    torch.cuda.synchronize()


def test_reshape_and_cache_flash(ops):
    seed_all(1)
    T, H, D, NB, BS = 13, 4, 64, 16, 16
    for dtype in (torch.half, torch.float):
        qkv = torch.randn(T, 3, H, D, dtype=dtype)
        kc = torch.randn(NB, BS, H, D, dtype=dtype)
        vc = torch.randn(NB, BS, H, D, dtype=dtype)
        slots = torch.tensor(random.sample(range(NB * BS), T), dtype=torch.long)
        kc_o, vc_o = kc.clone(), vc.clone()
        oracle.reshape_and_cache_flash(qkv[:, 1], qkv[:, 2], kc_o, vc_o, slots, "auto")
        kc_g, vc_g, qg = kc.to(DEV), vc.to(DEV), qkv.to(DEV)
        ops.reshape_and_cache_flash(qg[:, 1], qg[:, 2], kc_g, vc_g, slots.to(DEV), "auto")
        assert torch.equal(kc_g.cpu(), kc_o) and torch.equal(vc_g.cpu(), vc_o)
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache_flash(qg[:, 1], qg[:, 2], kc_g, vc_g, slots.to(DEV), "fp8")


def test_copy_blocks_golden(ops):
    g = load_golden("copy_blocks_f32")
    kcs = [torch.from_numpy(a.copy()).to(DEV) for a in g["k_in"]]
    vcs = [torch.from_numpy(a.copy()).to(DEV) for a in g["v_in"]]
    ops.copy_blocks(kcs, vcs, torch.from_numpy(g["mapping"]).to(DEV))
    for l in range(len(kcs)):
        assert np.array_equal(kcs[l].cpu().numpy(), g["k_out"][l])
        assert np.array_equal(vcs[l].cpu().numpy(), g["v_out"][l])


@pytest.mark.parametrize("num_mappings", [1, 256])
@pytest.mark.parametrize("num_layers", [1, 3])
@pytest.mark.parametrize("head_size,block_size", [(64, 8), (128, 16), (80, 32)])
@pytest.mark.parametrize("dtype,kv_cache_dtype", [(torch.half, "auto"), (torch.float, "auto"), (torch.half, "fp8")])
def test_copy_blocks(ops, num_mappings, num_layers, head_size, block_size, dtype, kv_cache_dtype):
    """tests/kernels/test_cache.py:47-111: distinct src / dst blocks, every layer's K and V."""
    seed_all(0)
    num_blocks, num_heads = 1024, 8
    assert 2 * num_mappings <= num_blocks
    src_blocks = random.sample(range(num_blocks), num_mappings)
    remaining = list(set(range(num_blocks)) - set(src_blocks))
    dst_blocks = random.sample(remaining, 2 * num_mappings)
    mapping = []
    for i in range(num_mappings):
        mapping.append((src_blocks[i], dst_blocks[2 * i]))
        mapping.append((src_blocks[i], dst_blocks[2 * i + 1]))
    kcs, vcs = create_kv_caches_with_random(num_blocks, block_size, num_layers, num_heads, head_size, kv_cache_dtype, dtype)
    k_o = [t.clone() for t in kcs]
    v_o = [t.clone() for t in vcs]
    bm = torch.tensor(mapping, dtype=torch.int64)
    oracle.copy_blocks(k_o, v_o, bm)
    k_g = [t.to(DEV) for t in kcs]
    v_g = [t.to(DEV) for t in vcs]
    ops.copy_blocks(k_g, v_g, bm.to(DEV))
    for a, b in zip(k_g + v_g, k_o + v_o):
        assert torch.equal(_int_view(a.cpu()), _int_view(b))


@pytest.mark.parametrize("direction", [("cuda", "cpu"), ("cpu", "cuda"), ("cuda", "cuda")])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_swap_blocks(ops, direction, kv_cache_dtype):
    """tests/kernels/test_cache.py:299-358"""
    seed_all(0)
    num_blocks, block_size, num_heads, head_size, num_mappings = 64, 16, 8, 64, 17
    src_dev = DEV if direction[0] == "cuda" else "cpu"
    dst_dev = DEV if direction[1] == "cuda" else "cpu"
    src_blocks = random.sample(range(num_blocks), num_mappings)
    if src_dev == dst_dev:
        remaining = list(set(range(num_blocks)) - set(src_blocks))
        dst_blocks = random.sample(remaining, num_mappings)
    else:
        dst_blocks = random.sample(range(num_blocks), num_mappings)
    bm = torch.tensor(list(zip(src_blocks, dst_blocks)), dtype=torch.int64)
    src_k, src_v = create_kv_caches_with_random(num_blocks, block_size, 1, num_heads, head_size, kv_cache_dtype, torch.half, seed=1)
    dst_k, dst_v = create_kv_caches_with_random(num_blocks, block_size, 1, num_heads, head_size, kv_cache_dtype, torch.half, seed=2)
    for src, dst in ((src_k[0], dst_k[0]), (src_v[0], dst_v[0])):
        dst_o = dst.clone()
        oracle.swap_blocks(src, dst_o, bm)
        src_t = src.to(src_dev) if src_dev != "cpu" else src.pin_memory()
        dst_t = dst.to(dst_dev) if dst_dev != "cpu" else dst.clone().pin_memory()
        ops.swap_blocks(src_t, dst_t, bm)
        torch.cuda.synchronize()
        assert torch.equal(_int_view(dst_t.cpu()), _int_view(dst_o))
    with pytest.raises(RuntimeError):
        ops.swap_blocks(src_k[0].to(DEV), dst_k[0].to(DEV), bm.to(DEV))  # block_mapping must be on CPU


@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("kv_dtype", ["fp8", "fp8_e5m2"])
def test_convert_fp8(ops, dtype, kv_dtype):
    """fp8 round trip (tests/kernels/test_cache.py:369-394) + exact agreement with the oracle's conversions."""
    seed_all(0)
    x = (torch.randn(4, 8, 16, 16, 8) * 3).to(dtype)
    x.view(-1)[:6] = torch.tensor([0.0, 448.0, 1e4, -1e4, 2**-9, 3e-3]).to(dtype)
    scale = 0.5
    q_o = torch.empty(x.shape, dtype=torch.uint8)
    oracle.convert_fp8(q_o, x, scale, kv_dtype)
    q_g = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    ops.convert_fp8(q_g, x.to(DEV), scale, kv_dtype)
    assert torch.equal(q_g.cpu(), q_o)
    back_o = torch.empty_like(x)
    oracle.convert_fp8(back_o, q_o, scale, kv_dtype)
    back_g = torch.empty_like(x, device=DEV)
    ops.convert_fp8(back_g, q_g, scale, kv_dtype)
    assert torch.equal(_int_view(back_g.cpu()), _int_view(back_o))
    torch.testing.assert_close(back_g.cpu().float().clamp(-200, 200), x.float().clamp(-200, 200), atol=1e-3, rtol=0.26)
