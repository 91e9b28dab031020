"""vins-mono_amd — MI355X (gfx950) implementation of VINS-Mono's two compute hot paths behind a C-ABI.

The product is ``lib/libvinsgpu.so`` (hand-written HIP kernels + C++ host code, see ``csrc/``) with
the interface of ``include/vinsgpu.h``.  The Python modules here are only the ctypes binding and the
synthetic-data generator used by ``tests/`` and ``bench.py``; there is NO CPU fallback: importing
``ba`` / ``fe`` raises if the shared library has not been built (``python __graft_entry__.py``).

The directory name contains a hyphen, so load it with ``__graft_entry__.load_package()`` (which
registers it as ``vins_mono_amd``) rather than a plain ``import``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvinsgpu.so")
_lib = None


def lib():
    """The loaded libvinsgpu.so (raises if it is missing — build it first)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built. "
                "Run `python __graft_entry__.py` (or `make -C vins-mono_amd/csrc`). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    return _lib
