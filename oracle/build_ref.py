"""TEST INFRASTRUCTURE ONLY — builds the reference's own CPU backend into ``oracle/_ref/``.

Recipe (verified in the build container): the reference's ``csrc/cpu/*.cpp`` are compiled
*where they lie* under ``/root/reference`` with ``torch.utils.cpp_extension.load`` and the flags of
``cmake/cpu_extension.cmake:15-17,53-57``. No reference source is copied into this repository; only the
resulting shared object lands in ``oracle/_ref/`` (git-ignored, but it travels to the GPU box).

The ops appear as ``torch.ops.nmref_cpu.paged_attention_v1/v2`` and
``torch.ops.nmref_cpu_cache_ops.{reshape_and_cache, copy_blocks}`` with the reference's op schema
(csrc/cpu/torch_bindings.cpp:13-37,101-107). Limits of the reference CPU backend: float32 / bfloat16 only,
block_size == 16, kv_scale == 1.0, no fp8 KV, no block-sparse, no swap_blocks.

If ``/root/reference`` is absent (the GPU box), ``build()`` is a no-op and ``load()`` uses a prebuilt
``oracle/_ref/nmref_cpu.so`` when present.
"""
import glob
import os
import sys

REF_ROOT = os.environ.get("NMX_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "nmref_cpu"
SOURCES = ["activation.cpp", "attention.cpp", "cache.cpp", "layernorm.cpp", "pos_encoding.cpp", "torch_bindings.cpp"]


def so_path() -> str:
    return os.path.join(OUT_DIR, NAME + ".so")


def build(verbose: bool = False) -> bool:
    """Returns True when oracle/_ref/nmref_cpu.so exists afterwards."""
    if os.path.exists(so_path()):
        return True
    csrc = os.path.join(REF_ROOT, "csrc")
    if not os.path.isdir(os.path.join(csrc, "cpu")):
        return False
    os.makedirs(OUT_DIR, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(
        name=NAME,
        sources=[os.path.join(csrc, "cpu", s) for s in SOURCES],
        extra_cflags=["-O2", "-fopenmp", "-DVLLM_CPU_EXTENSION", "-mavx512f", "-mavx512vl", "-mavx512bw",
                      "-mavx512dq", "-std=c++17"],
        extra_ldflags=["-fopenmp"],
        extra_include_paths=[csrc],
        build_directory=OUT_DIR,
        is_python_module=False,
        verbose=verbose,
    )
    # keep only the shared object
    for f in glob.glob(os.path.join(OUT_DIR, "*")):
        if not f.endswith(".so"):
            try:
                os.remove(f)
            except OSError:
                pass
    return os.path.exists(so_path())


_loaded = False


def load() -> bool:
    """Loads the prebuilt reference ops into torch.ops (True on success)."""
    global _loaded
    if _loaded:
        return True
    if not os.path.exists(so_path()):
        return False
    import torch
    torch.ops.load_library(so_path())
    _loaded = True
    return True


if __name__ == "__main__":
    ok = build(verbose="-v" in sys.argv)
    print("oracle/_ref built:" if ok else "oracle/_ref NOT built (reference absent)", so_path())
