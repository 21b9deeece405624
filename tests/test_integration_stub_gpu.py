"""The ctypes stub printed in INTEGRATION.md section 3 is executed as written (only the library path is substituted) and
must give the same results as the package's own binding - so the document cannot drift from include/nmx.h again."""
import os
import re

import pytest
import torch

from oracle import packing
from util import seed_all

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs(ops):
    from neuralmagic_vllm_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libnmx_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    seed_all(0)
    M, K, N = 48, 1024, 512
    w = torch.randn(K, N, dtype=torch.float16)
    _, mq, ms, _, _, _ = packing.marlin_quantize(w, 4, 128, False)
    a = torch.randn(M, K, dtype=torch.float16, device=DEV)
    e = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    c_stub = ns["gptq_marlin_gemm"](a, mq.to(DEV), ms.to(DEV), e, e, ws, 4, M, N, K, True)
    c_ops = ops.gptq_marlin_gemm(a, mq.to(DEV), ms.to(DEV), e, e, ws, 4, M, N, K, True)
    torch.cuda.synchronize()
    assert torch.equal(c_stub, c_ops)
    S, H, KVH, D, BS, NB, L = 3, 8, 2, 128, 16, 32, 100
    q = torch.randn(S, H, D, dtype=torch.float16, device=DEV) * 0.1
    kc = torch.randn(NB, KVH, D // 8, BS, 8, dtype=torch.float16, device=DEV) * 0.1
    vc = torch.randn(NB, KVH, D, BS, dtype=torch.float16, device=DEV) * 0.1
    bt = torch.randperm(NB, device=DEV)[:S * 7].reshape(S, 7).to(torch.int32)
    sl = torch.tensor([L, 33, 1], dtype=torch.int32, device=DEV)
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    ns["paged_attention_v1"](o1, q, kc, vc, KVH, D**-0.5, bt, sl, BS, L, None, "auto", 1.0)
    ops.paged_attention_v1(o2, q, kc, vc, KVH, D**-0.5, bt, sl, BS, L, None, "auto", 1.0)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
