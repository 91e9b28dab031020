import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def handle():
    """A vg_handle on cuda:0.  Fails loudly (no CPU fallback) if the HIP library is missing."""
    from vins_mono_amd import ba
    h = ba.Handle()
    yield h
    h.close()


# ---- CPU fiber emulation of the kernel sources (tests/simt): TEST INFRASTRUCTURE, see tests/simt/README.md ---------
SIMT_DIR = os.path.join(ROOT, "tests", "simt")
# (VINS_SIMT_LIB: another build of the emulated library, e.g. the AddressSanitizer one of `make -C tests/simt asan`)
SIMT_LIB = os.environ.get("VINS_SIMT_LIB") or os.path.join(SIMT_DIR, "_build", "libvinsgpu_simt.so")


def _build_simt():
    import fcntl
    import subprocess
    os.makedirs(os.path.join(SIMT_DIR, "_build"), exist_ok=True)
    # several processes may get here at once (xdist workers, the ranks of the world-2 tests): one make at a time
    with open(os.path.join(SIMT_DIR, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-C", SIMT_DIR, "-j", str(os.cpu_count() or 4)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the SIMT emulation library failed:\n" + r.stdout[-4000:])


def _simt_handle():
    import ctypes
    pkg = graft.load_package()
    _build_simt()
    saved = (pkg._lib, pkg.LIB_PATH)
    pkg._lib, pkg.LIB_PATH = ctypes.CDLL(SIMT_LIB, mode=ctypes.RTLD_LOCAL), SIMT_LIB
    try:
        from vins_mono_amd import ba
        h = ba.Handle()
    finally:
        pkg._lib, pkg.LIB_PATH = saved
    return h


@pytest.fixture(scope="session")
def simt_handle():
    """A vg_handle of the EMULATED library: the same kernel sources compiled for the CPU fiber emulator.  Used by the
    `not gpu` tests to check kernel logic (barriers, indexing, control flow) here, where there is no GPU."""
    h = _simt_handle()
    yield h
    h.close()


def new_handle():
    """A further handle of the kind the `handle` fixture gives (tests that drive two handles side by side)."""
    if os.environ.get("VINS_TEST_SIMT") == "1":
        return _simt_handle()
    from vins_mono_amd import ba
    return ba.Handle()


if os.environ.get("VINS_TEST_SIMT") == "1":
    # development switch: run the `-m gpu` parity tests against the emulated library (slow; BA-sized problems only)
    @pytest.fixture(scope="session")
    def handle():  # noqa: F811
        h = _simt_handle()
        yield h
        h.close()
