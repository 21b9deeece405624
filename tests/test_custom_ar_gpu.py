"""One-GPU loop-back test of the xGMI all-reduce kernels (one-stage and two-stage): nmx_custom_ar_loopback runs all N "ranks"
of one call as ONE launch (rank = blockIdx.y), so every workgroup of every rank is resident together and the mesh barriers
complete without relying on concurrently scheduled launches - one attempt, no retries, nothing skipped. Signal + scratch
blocks come from nmx_custom_ar_alloc_meta (uncached fine-grained memory, as in the product path); payloads are ordinary
allocations. It proves the epoch barriers, both schedules, the fixed-order fp32 sum (bitwise identical on every rank, and the
two-stage result equal to the one-stage one), the visibility of flags and scratch between workgroups on different XCDs, and
the timeout path. The real multi-GPU leg (IPC handles over xGMI) needs a node with >= 2 GPUs and stays behind NMX_CUSTOM_AR=1
until measured there."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
SPIN = 1 << 18  # ~ tens of milliseconds: a broken barrier fails the test quickly instead of stalling the GPU for seconds


class Meta:
    """world signal + scratch blocks from the product allocator."""

    def __init__(self, lib, world, scratch_bytes):
        lib.nmx_custom_ar_meta_size.restype = ctypes.c_int64
        self.lib, self.ptrs = lib, []
        self.sig_bytes = int(lib.nmx_custom_ar_meta_size())
        from neuralmagic_vllm_amd import _lib
        for _ in range(world):
            p = ctypes.c_void_p()
            _lib.check(lib.nmx_custom_ar_alloc_meta(ctypes.c_int64(self.sig_bytes + scratch_bytes), ctypes.byref(p)))
            self.ptrs.append(p)

    def array(self):
        return (ctypes.c_void_p * len(self.ptrs))(*[p.value for p in self.ptrs])

    def close(self):
        for p in self.ptrs:
            self.lib.nmx_custom_ar_free_meta(p)


def _loopback(world, dtype, numel, stages, rounds=3):
    from neuralmagic_vllm_amd import _lib
    from neuralmagic_vllm_amd.distributed.custom_all_reduce import custom_ar_scratch_bytes
    lib = _lib.lib()
    nbytes = numel * torch.empty(0, dtype=dtype).element_size()
    meta = Meta(lib, world, custom_ar_scratch_bytes(nbytes, world))
    bufs = [torch.empty(numel, dtype=dtype, device=DEV) for _ in range(world)]
    outs = [torch.empty(numel, dtype=dtype, device=DEV) for _ in range(world)]
    fas = []
    try:
        for r in range(world):  # handles only serve nmx_custom_ar_check here (the error word of rank r)
            fa = ctypes.c_void_p()
            _lib.check(lib.nmx_custom_ar_init(meta.array(), r, world, ctypes.c_int64(custom_ar_scratch_bytes(nbytes, world)), ctypes.byref(fa)))
            fas.append(fa)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for it in range(rounds):  # several rounds: the epoch counters must keep the barriers apart
            g = torch.Generator(device=DEV)
            g.manual_seed(it)
            for b in bufs:
                b.copy_(torch.randn(numel, device=DEV, generator=g).to(dtype))
            for o in outs:
                o.fill_(float("nan"))
            ref = torch.zeros(numel, dtype=torch.float32, device=DEV)
            for b in bufs:
                ref += b.float()  # rank order, fp32: what the kernels compute
            _lib.check(lib.nmx_custom_ar_loopback(meta.array(), (ctypes.c_void_p * world)(*[b.data_ptr() for b in bufs]),
                                                  (ctypes.c_void_p * world)(*[o.data_ptr() for o in outs]), world,
                                                  ctypes.c_int64(numel), _DT[dtype], stages, ctypes.c_uint32(SPIN), stream))
            torch.cuda.synchronize()
            for r in range(world):
                err = ctypes.c_int(-1)
                _lib.check(lib.nmx_custom_ar_check(fas[r], 1, ctypes.byref(err)))
                assert err.value == 0, f"rank {r} round {it}: a mesh barrier timed out"
            for r in range(world):
                assert torch.equal(outs[r], ref.to(dtype)), f"rank {r} round {it}"
                assert torch.equal(outs[r], outs[0])  # bitwise identical on every rank
    finally:
        for fa in fas:
            lib.nmx_custom_ar_dispose(fa)
        meta.close()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("numel", [8, 4096, 65536])
@pytest.mark.parametrize("stages", [1, 2])
def test_loopback_two_ranks(ops, dtype, numel, stages):
    _loopback(2, dtype, numel, stages)


@pytest.mark.parametrize("world", [4, 6, 8])
@pytest.mark.parametrize("stages", [1, 2])
def test_loopback_mesh(ops, world, stages):
    # 8 * 21 + 8 packets: the last rank's slice carries a remainder; 1 Mi elements: the decode message of configs[4] at batch 64
    for numel in (8 * 21 * world + 8, 64 * 8192):
        _loopback(world, torch.float16, numel, stages, rounds=2)


def test_default_schedule_rule(ops):
    """stages = 0 follows custom_all_reduce.cuh:442-450: [256, 8192] fp16 = 4 MiB at 8 ranks runs two-stage."""
    _loopback(8, torch.bfloat16, 256 * 8192, 0, rounds=1)
    _loopback(4, torch.float16, 8 * 8192, 0, rounds=1)


def test_barrier_timeout_sets_error_and_writes_nothing(ops):
    """A rank whose peer never arrives: bounded spin, error word set, output untouched (no sum of stale payloads)."""
    from neuralmagic_vllm_amd import _lib
    lib = _lib.lib()
    meta = Meta(lib, 2, 0)
    numel = 4096
    buf = torch.ones(numel, dtype=torch.float16, device=DEV)
    out = torch.full((numel, ), 7.0, dtype=torch.float16, device=DEV)
    fa = ctypes.c_void_p()
    try:
        _lib.check(lib.nmx_custom_ar_init(meta.array(), 0, 2, ctypes.c_int64(0), ctypes.byref(fa)))
        _lib.check(lib.nmx_custom_ar_set_spin_limit(fa, ctypes.c_uint32(1 << 12)))
        _lib.check(lib.nmx_custom_ar_register_buffer(fa, (ctypes.c_void_p * 2)(buf.data_ptr(), buf.data_ptr())))
        _lib.check(lib.nmx_custom_ar_all_reduce(fa, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(numel), 1,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))  # rank 1 is never launched
        torch.cuda.synchronize()
        err = ctypes.c_int(0)
        _lib.check(lib.nmx_custom_ar_check(fa, 1, ctypes.byref(err)))
        assert err.value != 0
        assert torch.equal(out, torch.full_like(out, 7.0))
        _lib.check(lib.nmx_custom_ar_check(fa, 0, ctypes.byref(err)))
        assert err.value == 0  # cleared
    finally:
        lib.nmx_custom_ar_dispose(fa)
        meta.close()


def test_unregistered_buffer_is_refused(ops):
    from neuralmagic_vllm_amd import _lib
    lib = _lib.lib()
    meta = Meta(lib, 2, 0)
    fa = ctypes.c_void_p()
    _lib.check(lib.nmx_custom_ar_init(meta.array(), 0, 2, ctypes.c_int64(0), ctypes.byref(fa)))
    x = torch.zeros(64, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="is not registered"):
        _lib.check(lib.nmx_custom_ar_all_reduce(fa, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(64), 1,
                                                ctypes.c_void_p(0)))
    with pytest.raises(RuntimeError, match="only supports num gpus"):
        _lib.check(lib.nmx_custom_ar_init(meta.array(), 0, 3, ctypes.c_int64(0), ctypes.byref(fa)))
    lib.nmx_custom_ar_dispose(fa)
    meta.close()
