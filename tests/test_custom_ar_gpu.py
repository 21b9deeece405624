"""One-GPU loop-back test of the xGMI one-shot all-reduce kernel: N "ranks" live in one process on cuda:0 (their signal
blocks and payload buffers are ordinary allocations, so no IPC is involved), one launch per rank on its own stream. It
proves the epoch barrier, the fixed-order fp32 sum and the bounded spins; the real multi-GPU leg (IPC handles over xGMI)
needs the driver's 8-GPU node and stays behind NMX_CUSTOM_AR=1 until measured there."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _run(world, dtype, numel, rounds=3):
    from neuralmagic_vllm_amd import _lib
    lib = _lib.lib()
    lib.nmx_custom_ar_meta_size.restype = ctypes.c_int64
    meta = [torch.zeros(int(lib.nmx_custom_ar_meta_size()), dtype=torch.uint8, device=DEV) for _ in range(world)]
    bufs = [torch.empty(numel, dtype=dtype, device=DEV) for _ in range(world)]
    outs = [torch.empty(numel, dtype=dtype, device=DEV) for _ in range(world)]
    fas = []
    for r in range(world):
        fa = ctypes.c_void_p()
        _lib.check(lib.nmx_custom_ar_init((ctypes.c_void_p * world)(*[m.data_ptr() for m in meta]), r, world, ctypes.byref(fa)))
        # registration order: pointer table is indexed by rank; the own pointer sits at [rank]
        _lib.check(lib.nmx_custom_ar_register_buffer(fa, (ctypes.c_void_p * world)(*[b.data_ptr() for b in bufs])))
        fas.append(fa)
    streams = [torch.cuda.Stream() for _ in range(world)]
    try:
        for it in range(rounds):  # several rounds: the epoch counters must keep the barriers apart
            g = torch.Generator(device=DEV)
            g.manual_seed(it)
            for b in bufs:
                b.copy_(torch.randn(numel, device=DEV, generator=g).to(dtype))
            torch.cuda.synchronize()
            ref = torch.zeros(numel, dtype=torch.float32, device=DEV)
            for b in bufs:
                ref += b.float()  # rank order, fp32: what the kernel computes
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    _lib.check(lib.nmx_custom_ar_all_reduce(fas[r], ctypes.c_void_p(bufs[r].data_ptr()), ctypes.c_void_p(outs[r].data_ptr()),
                                                            ctypes.c_int64(numel), _DT[dtype],
                                                            ctypes.c_void_p(streams[r].cuda_stream)))
            torch.cuda.synchronize()
            errs = [int(m[-128:].view(torch.int32).abs().max()) for m in meta]
            if any(errs):
                return "timeout"  # the launches did not run concurrently (shared hardware queue): bounded spins gave up
            for r in range(world):
                assert torch.equal(outs[r], ref.to(dtype)), f"rank {r} round {it}"
                assert torch.equal(outs[r], outs[0])  # bitwise identical on every rank
    finally:
        for fa in fas:
            lib.nmx_custom_ar_dispose(fa)
    return "ok"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("numel", [8, 4096, 65536])
def test_loopback_two_ranks(ops, dtype, numel):
    _loopback(2, dtype, numel)


def _loopback(world, dtype, numel):
    # Streams of ONE process may share a hardware queue; two launches that share one run back to back, the first spins
    # until its bound and reports through the error word (no hang). That is an artefact of the loop-back arrangement -
    # real ranks are separate processes on separate GPUs - so such a round is retried on fresh streams, then skipped.
    for _ in range(3):
        if _run(world, dtype, numel) == "ok":
            return
    pytest.skip(f"{world} concurrent launches of one process did not get separate hardware queues")


def test_loopback_four_ranks(ops):
    _loopback(4, torch.float16, 32768)


def test_unregistered_buffer_is_refused(ops):
    from neuralmagic_vllm_amd import _lib
    lib = _lib.lib()
    lib.nmx_custom_ar_meta_size.restype = ctypes.c_int64
    meta = [torch.zeros(int(lib.nmx_custom_ar_meta_size()), dtype=torch.uint8, device=DEV) for _ in range(2)]
    fa = ctypes.c_void_p()
    _lib.check(lib.nmx_custom_ar_init((ctypes.c_void_p * 2)(*[m.data_ptr() for m in meta]), 0, 2, ctypes.byref(fa)))
    x = torch.zeros(64, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="is not registered"):
        _lib.check(lib.nmx_custom_ar_all_reduce(fa, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(64), 1,
                                                ctypes.c_void_p(0)))
    with pytest.raises(RuntimeError, match="only supports num gpus"):
        _lib.check(lib.nmx_custom_ar_init((ctypes.c_void_p * 2)(*[m.data_ptr() for m in meta]), 0, 3, ctypes.byref(fa)))
    lib.nmx_custom_ar_dispose(fa)
