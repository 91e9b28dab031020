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


def _prebuild_for_workers():
    """Everything a `not gpu` test would (re)build on demand, built ONCE before the worker processes start: afterwards their own `make`
    calls find nothing to do, and a make that finds nothing to do may run next to another one."""
    import shutil
    import subprocess
    jobs = str(os.cpu_count() or 4)
    _build_simt()
    for args, cwd in ((["make", "-j", jobs], os.path.join(ROOT, "oracle")),):
        subprocess.run(args, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "ref_simt"], cwd=os.path.join(ROOT, "oracle"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True) if shutil.which("gcc") else None
    if r is not None and r.returncode == 0 and os.path.isabs(r.stdout.strip()) and os.path.exists(r.stdout.strip()):
        # (outside the repository, where test_simt_asan.py / test_abi_fuzz.py look for it: the GPU pool refuses repository snapshots that carry
        #  -fsanitize=address objects)
        import tempfile
        asan_out = os.path.join(tempfile.gettempdir(), "vins_simt_build_asan_%d" % os.getuid())
        subprocess.run(["make", "-C", SIMT_DIR, "-j", jobs, "asan", "ASAN_OUT=" + asan_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest tests -m "not gpu"` spreads the test FILES over worker processes (pytest-xdist, --dist loadfile).  The CPU suite is a dozen
    independent, subprocess-heavy tests long (the emulated bench, the end-to-end runs on rendered frames, the sanitizer builds): 14
    minutes in sequence, a few in parallel.  Files stay whole on one worker (module fixtures, fixed rendezvous ports per file).  Not
    for the GPU suite (one device), not when the caller chose a worker count or VINS_TEST_SERIAL=1 is set."""
    opt = config.option
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or os.environ.get("VINS_TEST_SERIAL") == "1":
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) not in (None, 0) or getattr(opt, "collectonly", False):
        return None
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    _prebuild_for_workers()
    opt.numprocesses = max(2, min(8, (os.cpu_count() or 4) * 3 // 4))
    opt.dist = "loadfile"
    config._vins_heavy_first = True
    return None


# the long files of the `not gpu` suite, longest first (seconds under load: 154, 134, 88, 86, 58, 55, 47, 37): handed to the workers
# before everything else, so that none of them starts when the others are nearly done
_HEAVY_FILES = ("test_e2e_simt.py", "test_simt_asan.py", "test_bench_selflaunch.py", "test_bench_contract.py", "test_fe_read_image.py",
                "test_fe_dropin.py", "test_seq_simt.py", "test_simt_ba.py")


def pytest_collection_modifyitems(config, items):
    if not (getattr(config, "_vins_heavy_first", False) or os.environ.get("PYTEST_XDIST_WORKER")):
        return
    rank = {f: i for i, f in enumerate(_HEAVY_FILES)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(rank)))       # (stable: the rest keeps its order)


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
