"""Element-wise ops: oracle vs the reference CPU backend's golden outputs (CPU), HIP vs oracle (GPU).
Mirrors tests/kernels/test_layernorm.py, test_pos_encoding.py, test_activation.py of the reference."""
import pytest
import torch

import oracle
from util import DTYPES, from_bits, load_golden, seed_all

DEV = "cuda:0"


def _tol(dt):
    # reference tests: layernorm atol 1e-2 rtol 1e-2 (test_layernorm.py), activation 1e-5 (fp32)
    return dict(atol=2e-5, rtol=1e-5) if dt == torch.float32 else dict(atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_oracle_vs_reference_cpu_backend(tag):
    g = load_golden("elementwise")
    dt = DTYPES[tag]
    x, res, w = (from_bits(g[f"rms_{n}_{tag}"], dt) for n in ("x", "res", "w"))
    out = torch.empty_like(x)
    oracle.rms_norm(out, x, w, 1e-5)
    torch.testing.assert_close(out.float(), from_bits(g[f"rms_out_{tag}"], dt).float(), **_tol(dt))
    x2, r2 = x.clone(), res.clone()
    oracle.fused_add_rms_norm(x2, r2, w, 1e-5)
    torch.testing.assert_close(x2.float(), from_bits(g[f"fused_out_{tag}"], dt).float(), **_tol(dt))
    torch.testing.assert_close(r2.float(), from_bits(g[f"fused_res_{tag}"], dt).float(), **_tol(dt))
    for name, neox in (("neox", True), ("gptj", False)):
        key = f"rope_{name}_{tag}"
        q, k, cache = (from_bits(g[f"{key}_{n}"], dt) for n in ("q", "k", "cache"))
        oracle.rotary_embedding(torch.from_numpy(g[key + "_pos"]), q, k, 64, cache, neox)
        torch.testing.assert_close(q.float(), from_bits(g[key + "_qo"], dt).float(), **_tol(dt))
        torch.testing.assert_close(k.float(), from_bits(g[key + "_ko"], dt).float(), **_tol(dt))
    gin = from_bits(g[f"gate_in_{tag}"], dt)
    for name, kind in (("silu_and_mul", "silu"), ("gelu_and_mul", "gelu"), ("gelu_tanh_and_mul", "gelu_tanh")):
        o = torch.empty(gin.shape[0], gin.shape[1] // 2, dtype=dt)
        oracle.act_and_mul(o, gin, kind)
        torch.testing.assert_close(o.float(), from_bits(g[f"{name}_{tag}"], dt).float(), **_tol(dt))
    a = from_bits(g[f"act_in_{tag}"], dt)
    for name in ("gelu_new", "gelu_fast"):
        o = torch.empty_like(a)
        oracle.activation(o, a, name)
        torch.testing.assert_close(o.float(), from_bits(g[f"{name}_{tag}"], dt).float(), **_tol(dt))


def _bits(t):
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16)


def _close_to_oracle(got, want, dt, transcendental=False):
    """Pure arithmetic (norm, rope) is bit-exact for 16-bit types; device exp/tanh/erf differ from libm by an ulp of
    fp32, which can flip the last bit of the 16-bit result."""
    if dt != torch.float32 and not transcendental:
        assert torch.equal(_bits(got), _bits(want))
    else:
        ulp = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dt]
        torch.testing.assert_close(got.float(), want.float(), atol=ulp, rtol=ulp)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("num_tokens,hidden", [(1, 4096), (7, 768), (83, 8192), (5, 8199), (3, 16384)])
def test_rms_norm(ops, dtype, num_tokens, hidden):
    seed_all(0)
    x = torch.randn(num_tokens, hidden, dtype=dtype)
    res = torch.randn(num_tokens, hidden, dtype=dtype)
    w = torch.randn(hidden, dtype=dtype)
    want = torch.empty_like(x)
    oracle.rms_norm(want, x, w, 1e-6)
    got = torch.empty_like(x, device=DEV)
    ops.rms_norm(got, x.to(DEV), w.to(DEV), 1e-6)
    # fp32 accumulation order differs (tree vs serial): compare to rounding, not bits
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol, rtol=tol)
    x2, r2 = x.clone(), res.clone()
    oracle.fused_add_rms_norm(x2, r2, w, 1e-6)
    xg, rg = x.to(DEV), res.to(DEV)
    ops.fused_add_rms_norm(xg, rg, w.to(DEV), 1e-6)
    assert torch.equal(_bits(rg.cpu()), _bits(r2))  # the residual add is exact arithmetic
    torch.testing.assert_close(xg.cpu().float(), x2.float(), atol=tol, rtol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("head_size,rot_dim", [(128, 128), (64, 32), (80, 80)])
def test_rotary_embedding(ops, dtype, is_neox, head_size, rot_dim):
    seed_all(1)
    T, nh, nkv, max_pos = 11, 8, 2, 512
    pos = torch.randint(0, max_pos, (T, ), dtype=torch.int64)
    qkv = torch.randn(T, (nh + 2 * nkv) * head_size, dtype=dtype)
    cache = torch.randn(max_pos, rot_dim, dtype=dtype)
    q_o = qkv[:, :nh * head_size].clone()
    k_o = qkv[:, nh * head_size:(nh + nkv) * head_size].clone()
    oracle.rotary_embedding(pos, q_o, k_o, head_size, cache, is_neox)
    g = qkv.to(DEV)
    q_g = g[:, :nh * head_size]  # strided views into the fused qkv row, like the reference model code
    k_g = g[:, nh * head_size:(nh + nkv) * head_size]
    ops.rotary_embedding(pos.to(DEV), q_g, k_g, head_size, cache.to(DEV), is_neox)
    # hipcc contracts x*c - y*s into mixed-precision FMAs (fewer intermediate roundings than the reference's
    # scalar_t-by-scalar_t arithmetic): results agree to one ulp of the storage type, not always bit for bit
    ulp = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    torch.testing.assert_close(q_g.cpu().float(), q_o.float(), atol=ulp, rtol=ulp)
    torch.testing.assert_close(k_g.cpu().float(), k_o.float(), atol=ulp, rtol=ulp)
    assert torch.equal(g[:, (nh + nkv) * head_size:].cpu(), qkv[:, (nh + nkv) * head_size:])  # v untouched
    # batched variant with per-token cache offsets
    offs = torch.randint(0, 4, (T, ), dtype=torch.int64) * 64
    cache2 = torch.randn(max_pos + 256, rot_dim, dtype=dtype)
    q_o2, k_o2 = qkv[:, :nh * head_size].clone(), qkv[:, nh * head_size:(nh + nkv) * head_size].clone()
    oracle.rotary_embedding(pos, q_o2, k_o2, head_size, cache2, is_neox, offs)
    q_g2, k_g2 = q_o2.clone().zero_().to(DEV), None
    q_g2 = qkv[:, :nh * head_size].clone().to(DEV)
    k_g2 = qkv[:, nh * head_size:(nh + nkv) * head_size].clone().to(DEV)
    ops.batched_rotary_embedding(pos.to(DEV), q_g2, k_g2, head_size, cache2.to(DEV), is_neox, rot_dim, offs.to(DEV))
    torch.testing.assert_close(q_g2.cpu().float(), q_o2.float(), atol=ulp, rtol=ulp)
    torch.testing.assert_close(k_g2.cpu().float(), k_o2.float(), atol=ulp, rtol=ulp)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.half, torch.bfloat16, torch.float])
@pytest.mark.parametrize("num_tokens,d", [(1, 14336), (7, 512), (83, 13824), (5, 33)])
def test_activations(ops, dtype, num_tokens, d):
    seed_all(2)
    x = torch.randn(num_tokens, 2 * d, dtype=dtype)
    for fn, kind in (("silu_and_mul", "silu"), ("gelu_and_mul", "gelu"), ("gelu_tanh_and_mul", "gelu_tanh")):
        want = torch.empty(num_tokens, d, dtype=dtype)
        oracle.act_and_mul(want, x, kind)
        got = torch.empty(num_tokens, d, dtype=dtype, device=DEV)
        getattr(ops, fn)(got, x.to(DEV))
        _close_to_oracle(got.cpu(), want, dtype, transcendental=True)
    a = torch.randn(num_tokens, d, dtype=dtype)
    for fn in ("gelu_new", "gelu_fast", "gelu_quick"):
        want = torch.empty_like(a)
        oracle.activation(want, a, fn)
        got = torch.empty_like(a, device=DEV)
        getattr(ops, fn)(got, a.to(DEV))
        _close_to_oracle(got.cpu(), want, dtype, transcendental=True)
