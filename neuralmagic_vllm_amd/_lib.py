"""ctypes loader for ``libnmx_hip.so`` (the C-ABI in ``include/nmx.h``).

The product path fails loudly when the HIP library is missing: there is deliberately no fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMX_LIB_PATH: load another build of the same HIP library (kernel experiments); never a non-HIP substitute
LIB_PATH = os.environ.get("NMX_LIB_PATH") or os.path.join(_HERE, "libnmx_hip.so")

_lib = None


class NmxLibraryMissing(ImportError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NmxLibraryMissing(
                f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for these ops.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.nmx_last_error.restype = ctypes.c_char_p
        _lib.nmx_version.restype = ctypes.c_char_p
        for fn in ("nmx_marlin_gemm_scratch_bytes", "nmx_zp_gemm_scratch_bytes", "nmx_scaled_mm_scratch_bytes"):
            getattr(_lib, fn).restype = ctypes.c_int64
        _lib.nmx_tuning_set.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    return _lib


def set_tuning(name: str, value=None) -> None:
    """Kernel sweeps / tests: override (or clear, value=None) one of the NMX_* tuning variables after load."""
    check(lib().nmx_tuning_set(name.encode(), None if value is None else str(value).encode()))


def check(rc: int) -> None:
    """Maps a negative NMX_ERR_* code to RuntimeError (reference: TORCH_CHECK -> c10::Error -> RuntimeError)."""
    if rc != 0:
        msg = lib().nmx_last_error()
        raise RuntimeError(msg.decode() if msg else f"nmx error {rc}")
