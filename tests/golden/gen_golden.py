"""Generates the golden fixtures in tests/golden/ from the REFERENCE itself (run in the build container only).

Two sources, both read-only under /root/reference, neither copied into this repository:
  1. the reference's CPU backend (csrc/cpu/*.cpp) compiled into oracle/_ref by oracle/build_ref.py
     -> paged_attention_v1/v2, reshape_and_cache, copy_blocks outputs (float32 / bfloat16, block_size 16);
  2. the reference's Python quantization utilities (vllm/model_executor/layers/quantization/utils/*.py), imported
     through namespace-stub packages so that vllm/__init__.py (which needs absent dependencies) never executes
     -> quantize_weights / gptq_pack / marlin_weights / marlin_permute_scales / sort_weights / 2:4 helpers.

Fixtures are plain data (inputs + expected outputs) stored as .npz; bf16 / fp16 tensors are stored as uint16 bit
patterns. Usage:  python tests/golden/gen_golden.py
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("NMX_REFERENCE_ROOT", "/root/reference")


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


# ------------------------------------------------------------------------------------------------------------
# 1. reference CPU backend
# ------------------------------------------------------------------------------------------------------------
def gen_attention_cache():
    from oracle import build_ref
    assert build_ref.build(), "reference CPU backend could not be built"
    assert build_ref.load()
    ops = torch.ops.nmref_cpu
    cops = torch.ops.nmref_cpu_cache_ops

    def make_case(seed, dtype, S, H, KVH, D, NB, max_len, alibi):
        torch.manual_seed(seed)
        random.seed(seed)
        BS = 16
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        scale = float(D**-0.5)
        q = torch.empty(S, H, D, dtype=dtype).uniform_(-scale, scale)
        kc = torch.empty(NB, KVH, D // x, BS, x, dtype=dtype).uniform_(-scale, scale)
        vc = torch.empty(NB, KVH, D, BS, dtype=dtype).uniform_(-scale, scale)
        seq_lens = [random.randint(1, max_len) for _ in range(S)]
        seq_lens[-1] = max_len
        mb = (max_len + BS - 1) // BS
        bt = torch.tensor([[random.randint(0, NB - 1) for _ in range(mb)] for _ in range(S)], dtype=torch.int32)
        sl = torch.tensor(seq_lens, dtype=torch.int32)
        al = torch.randn(H, dtype=torch.float32) if alibi else None
        out1 = torch.empty_like(q)
        ops.paged_attention_v1(out1, q, kc, vc, KVH, scale, bt, sl, BS, max_len, al, "auto", 1.0, 0, 0, 0, 64, 0)
        P = (max_len + 511) // 512
        out2 = torch.empty_like(q)
        tmp = torch.empty(S, H, P, D, dtype=dtype)
        es = torch.empty(S, H, P, dtype=torch.float32)
        ml = torch.empty(S, H, P, dtype=torch.float32)
        ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, BS, max_len, al, "auto", 1.0, 0, 0,
                               0, 64, 0)
        d = dict(q=bits(q), k_cache=bits(kc), v_cache=bits(vc), block_tables=bt.numpy(), seq_lens=sl.numpy(),
                 out_v1=bits(out1), out_v2=bits(out2), scale=np.float32(scale), num_kv_heads=np.int32(KVH),
                 max_seq_len=np.int32(max_len), dtype=str(dtype).split(".")[-1])
        if al is not None:
            d["alibi_slopes"] = al.numpy()
        return d

    cases = {
        "attn_f32_gqa": make_case(0, torch.float32, 3, 8, 2, 64, 24, 600, False),
        "attn_bf16_gqa": make_case(1, torch.bfloat16, 4, 8, 2, 128, 24, 700, False),
        "attn_bf16_alibi_mha": make_case(2, torch.bfloat16, 3, 4, 4, 64, 20, 150, True),
    }
    for name, d in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print("wrote", name)

    # reshape_and_cache + copy_blocks
    torch.manual_seed(3)
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        T, H, D, NB, BS = 7, 4, 64, 10, 16
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        qkv = torch.randn(T, 3, H, D, dtype=dtype)
        key, value = qkv[:, 1], qkv[:, 2]  # strided views like the reference test (test_cache.py:150-151)
        kc = torch.randn(NB, H, D // x, BS, x, dtype=dtype)
        vc = torch.randn(NB, H, D, BS, dtype=dtype)
        slots = torch.tensor(random.sample(range(NB * BS), T), dtype=torch.int64)
        slots[2] = -1  # padding token
        kc_in, vc_in = kc.clone(), vc.clone()
        cops.reshape_and_cache(key, value, kc, vc, slots, "auto", 1.0)
        np.savez_compressed(os.path.join(HERE, f"reshape_and_cache_{tag}.npz"), qkv=bits(qkv), k_cache_in=bits(kc_in),
                            v_cache_in=bits(vc_in), slot_mapping=slots.numpy(), k_cache_out=bits(kc),
                            v_cache_out=bits(vc), dtype=tag)
        print("wrote reshape_and_cache", tag)

    torch.manual_seed(4)
    L, NB, KVH, D, BS = 3, 12, 2, 32, 16
    kcs = [torch.randn(NB, KVH, D // 4, BS, 4) for _ in range(L)]
    vcs = [torch.randn(NB, KVH, D, BS) for _ in range(L)]
    mapping = torch.tensor([[0, 5], [0, 7], [3, 9], [4, 10]], dtype=torch.int64)
    k_in = np.stack([t.numpy().copy() for t in kcs])
    v_in = np.stack([t.numpy().copy() for t in vcs])
    cops.copy_blocks(kcs, vcs, mapping)
    np.savez_compressed(os.path.join(HERE, "copy_blocks_f32.npz"), k_in=k_in, v_in=v_in, mapping=mapping.numpy(),
                        k_out=np.stack([t.numpy() for t in kcs]), v_out=np.stack([t.numpy() for t in vcs]))
    print("wrote copy_blocks")


# ------------------------------------------------------------------------------------------------------------
# 2. reference Python quantization utilities
# ------------------------------------------------------------------------------------------------------------
def import_ref_utils():
    R = os.path.join(REF, "vllm")
    for name, path in [
        ("vllm", R),
        ("vllm.model_executor", R + "/model_executor"),
        ("vllm.model_executor.layers", R + "/model_executor/layers"),
        ("vllm.model_executor.layers.quantization", R + "/model_executor/layers/quantization"),
        ("vllm.model_executor.layers.quantization.utils", R + "/model_executor/layers/quantization/utils"),
    ]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    plat = types.ModuleType("vllm.platforms")  # marlin_utils imports current_platform only for is_marlin_supported()
    plat.current_platform = types.SimpleNamespace(get_device_capability=lambda: (9, 5))
    sys.modules["vllm.platforms"] = plat
    from vllm.model_executor.layers.quantization.utils import (format_24, marlin_24_perms, marlin_perms,  # noqa
                                                               marlin_utils, quant_utils)
    return quant_utils, marlin_perms, marlin_utils, format_24, marlin_24_perms


def gen_quant():
    qu, mp, mu, f24, mp24 = import_ref_utils()
    for bits_, gs, act in [(4, 128, False), (4, -1, False), (4, 64, True), (8, 128, False), (8, -1, False),
                           (8, 32, True)]:
        torch.manual_seed(100 + bits_ + (gs if gs > 0 else 7) + int(act))
        K, N = 256, 192
        w = torch.randn(K, N, dtype=torch.float16)
        g = K if gs == -1 else gs
        w_ref, q_w, s, g_idx, rand_perm = qu.quantize_weights(w, bits_, g, act)
        q_gptq = qu.gptq_pack(q_w, bits_, K, N)
        sort_idx = torch.empty(0, dtype=torch.int32)
        q_sorted, g_sorted = q_w, g_idx
        if act:
            q_sorted, g_sorted, sort_idx = qu.sort_weights(q_w, g_idx)
        marlin_q = mu.marlin_weights(q_sorted, K, N, bits_, mp.marlin_perm[bits_])
        marlin_s = mu.marlin_permute_scales(s, K, N, g, mp.marlin_scale_perm[bits_], mp.marlin_scale_perm_single[bits_])
        torch.manual_seed(7)
        a = torch.randn(5, K, dtype=torch.float16)
        c_ref = torch.matmul(a.float(), w_ref.float())
        name = f"marlin_b{bits_}_g{gs}_act{int(act)}"
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), w=bits(w), w_ref=bits(w_ref), q_w=q_w.numpy().astype(np.uint8),
            s=bits(s), g_idx=g_idx.numpy().astype(np.int32), rand_perm=rand_perm.numpy().astype(np.int64),
            q_gptq=q_gptq.numpy(), sort_idx=sort_idx.numpy().astype(np.int32),
            g_idx_sorted=g_sorted.numpy().astype(np.int32), marlin_q=marlin_q.numpy(), marlin_s=bits(marlin_s),
            a=bits(a), c_ref=c_ref.numpy(), bits=np.int32(bits_), group_size=np.int32(gs))
        print("wrote", name)

    # fp8 byte packing (marlin_utils.pack_fp8_to_int32)
    torch.manual_seed(11)
    w8 = torch.randn(64, 128).to(torch.float8_e4m3fn)
    packed = mu.pack_fp8_to_int32(w8)
    np.savez_compressed(os.path.join(HERE, "pack_fp8.npz"), w8=w8.view(torch.uint8).numpy(), packed=packed.numpy())
    print("wrote pack_fp8")

    # 2:4 sparse Marlin pieces (marlin_24_quantize needs .cuda(); its pieces do not — marlin_utils.py:180-198)
    for bits_, gs in [(4, 128), (4, -1), (8, 128)]:
        torch.manual_seed(200 + bits_ + (gs if gs > 0 else 3))
        K, N = 256, 256
        w = torch.randn(K, N, dtype=torch.float16)
        g = K if gs == -1 else gs
        mask = f24.mask_creator(w.t()).t().bool()
        w_24 = (mask * w).contiguous()
        w_24_ref, q_w_24, s, _, _ = qu.quantize_weights(w_24, bits_, g, False)
        q_comp, meta = mu.compress_quantized_24_weight(q_w_24, K, N, bits_)
        marlin_24_q = mu.marlin_weights(q_comp, K // 2, N, bits_, mp24.marlin_24_perm[bits_])
        marlin_24_s = mu.marlin_permute_scales(s, K, N, g, mp24.marlin_24_scale_perm[bits_],
                                               mp24.marlin_24_scale_perm_single[bits_])
        torch.manual_seed(8)
        a = torch.randn(5, K, dtype=torch.float16)
        c_ref = torch.matmul(a.float(), w_24_ref.float())
        name = f"marlin24_b{bits_}_g{gs}"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), w=bits(w), mask=mask.numpy(), w_24_ref=bits(w_24_ref),
                            q_w_24=q_w_24.numpy().astype(np.uint8), s=bits(s), q_comp=q_comp.numpy().astype(np.uint8),
                            meta=meta.numpy(), marlin_24_q=marlin_24_q.numpy(), marlin_24_s=bits(marlin_24_s),
                            a=bits(a), c_ref=c_ref.numpy(), bits=np.int32(bits_), group_size=np.int32(gs))
        print("wrote", name)


def gen_elementwise():
    """rms_norm / fused_add_rms_norm / rotary_embedding / gated activations from the reference CPU backend
    (csrc/cpu/{layernorm,pos_encoding,activation}.cpp; schema csrc/cpu/torch_bindings.cpp:39-99)."""
    from oracle import build_ref
    assert build_ref.build() and build_ref.load()
    ops = torch.ops.nmref_cpu
    torch.manual_seed(5)
    d = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        T, H = 5, 512
        x = torch.randn(T, H, dtype=dtype)
        res = torch.randn(T, H, dtype=dtype)
        w = torch.randn(H, dtype=dtype)
        out = torch.empty_like(x)
        ops.rms_norm(out, x, w, 1e-5)
        x2, r2 = x.clone(), res.clone()
        ops.fused_add_rms_norm(x2, r2, w, 1e-5)
        d.update({f"rms_x_{tag}": bits(x), f"rms_res_{tag}": bits(res), f"rms_w_{tag}": bits(w), f"rms_out_{tag}": bits(out),
                  f"fused_out_{tag}": bits(x2), f"fused_res_{tag}": bits(r2)})
        # rotary: 6 heads / 2 kv heads x 64, rot_dim 64 (neox) and 32 (gptj, partial)
        for neox, rot in ((True, 64), (False, 32)):
            pos = torch.tensor([0, 3, 17, 100, 255], dtype=torch.int64)
            q = torch.randn(T, 6 * 64, dtype=dtype)
            k = torch.randn(T, 2 * 64, dtype=dtype)
            cache = torch.randn(256, rot, dtype=dtype)
            q2, k2 = q.clone(), k.clone()
            ops.rotary_embedding(pos, q2, k2, 64, cache, neox)
            key = f"rope_{'neox' if neox else 'gptj'}_{tag}"
            d.update({key + "_pos": pos.numpy(), key + "_q": bits(q), key + "_k": bits(k), key + "_cache": bits(cache),
                      key + "_qo": bits(q2), key + "_ko": bits(k2)})
        g = torch.randn(T, 2 * 192, dtype=dtype)
        for name in ("silu_and_mul", "gelu_and_mul", "gelu_tanh_and_mul"):
            o = torch.empty(T, 192, dtype=dtype)
            getattr(ops, name)(o, g)
            d[f"{name}_{tag}"] = bits(o)
        d[f"gate_in_{tag}"] = bits(g)
        a = torch.randn(T, 192, dtype=dtype)
        for name in ("gelu_new", "gelu_fast"):
            o = torch.empty_like(a)
            getattr(ops, name)(o, a)
            d[f"{name}_{tag}"] = bits(o)
        d[f"act_in_{tag}"] = bits(a)
    np.savez_compressed(os.path.join(HERE, "elementwise.npz"), **d)
    print("wrote elementwise")


def gen_opt125m():
    """BASELINE configs[0] geometry (OPT-125m: 12 MHA heads x 64, hidden 768, block 16, bf16 - the dtype the reference's
    CPU backend decodes it in): paged_attention_v1 / v2 and reshape_and_cache of the reference CPU backend
    (csrc/cpu/attention.cpp:222-439, csrc/cpu/cache.cpp)."""
    from oracle import build_ref
    assert build_ref.build() and build_ref.load()
    ops, cops = torch.ops.nmref_cpu, torch.ops.nmref_cpu_cache_ops
    torch.manual_seed(125)
    random.seed(125)
    dtype, S, H, D, NB, BS, max_len = torch.bfloat16, 5, 12, 64, 20, 16, 600
    scale = float(D**-0.5)
    q = torch.empty(S, H, D, dtype=dtype).uniform_(-scale, scale)
    kc = torch.empty(NB, H, D // 8, BS, 8, dtype=dtype).uniform_(-scale, scale)
    vc = torch.empty(NB, H, D, BS, dtype=dtype).uniform_(-scale, scale)
    seq_lens = [1, 16, 513, 130, max_len]
    mb = (max_len + BS - 1) // BS
    bt = torch.tensor([[random.randint(0, NB - 1) for _ in range(mb)] for _ in range(S)], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    out1, out2 = torch.empty_like(q), torch.empty_like(q)
    ops.paged_attention_v1(out1, q, kc, vc, H, scale, bt, sl, BS, max_len, None, "auto", 1.0, 0, 0, 0, 64, 0)
    P = (max_len + 511) // 512
    tmp = torch.empty(S, H, P, D, dtype=dtype)
    es, ml = torch.empty(S, H, P, dtype=torch.float32), torch.empty(S, H, P, dtype=torch.float32)
    ops.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, H, scale, bt, sl, BS, max_len, None, "auto", 1.0, 0, 0, 0, 64, 0)
    np.savez_compressed(os.path.join(HERE, "attn_bf16_opt125m.npz"), q=bits(q), k_cache=bits(kc), v_cache=bits(vc),
                        block_tables=bt.numpy(), seq_lens=sl.numpy(), out_v1=bits(out1), out_v2=bits(out2),
                        scale=np.float32(scale), num_kv_heads=np.int32(H), max_seq_len=np.int32(max_len), dtype="bfloat16")
    T = 9
    qkv = torch.randn(T, 3, H, D, dtype=dtype)
    NB2 = 4
    kc2, vc2 = torch.randn(NB2, H, D // 8, BS, 8, dtype=dtype), torch.randn(NB2, H, D, BS, dtype=dtype)
    slots = torch.tensor(random.sample(range(NB2 * BS), T), dtype=torch.int64)
    slots[4] = -1
    k_in, v_in = kc2.clone(), vc2.clone()
    cops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc2, vc2, slots, "auto", 1.0)
    np.savez_compressed(os.path.join(HERE, "reshape_and_cache_bf16_opt125m.npz"), qkv=bits(qkv), k_cache_in=bits(k_in),
                        v_cache_in=bits(v_in), slot_mapping=slots.numpy(), k_cache_out=bits(kc2), v_cache_out=bits(vc2),
                        dtype="bf16")
    print("wrote attn_bf16_opt125m, reshape_and_cache_bf16_opt125m")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "elementwise":
        gen_elementwise()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "opt125m":
        gen_opt125m()
        sys.exit(0)
    gen_quant()
    gen_attention_cache()
    gen_opt125m()
    gen_elementwise()
