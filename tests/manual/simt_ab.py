"""Bitwise A/B of two builds of the library (normally two CPU-emulated builds, tests/simt/_build/libvinsgpu_simt.so of two
source trees): the same windows through both, every output compared bit for bit.  For kernel changes that re-arrange memory
traffic or launches but must not change a single rounding:  python tests/manual/simt_ab.py <libA.so> <libB.so>"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
from vins_mono_amd import ba, synth  # noqa: E402


def handle_of(path):
    saved = (pkg._lib, pkg.LIB_PATH)
    pkg._lib, pkg.LIB_PATH = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), path
    try:
        return ba.Handle()
    finally:
        pkg._lib, pkg.LIB_PATH = saved


def cases():
    yield "K=11 L=30 MARGIN_OLD", [synth.SyntheticSequence(3, L=30).window(0)], [ba.VG_MARGIN_OLD]
    yield "K=5 ex+td", [synth.SyntheticSequence(73, n_frames=6, K=5, L=20, estimate_extrinsic=1, estimate_td=1).window(0)], [ba.VG_MARGIN_OLD]
    yield "two windows", [synth.SyntheticSequence(11 + s, L=24).window(0) for s in range(2)], [ba.VG_MARGIN_OLD, ba.VG_MARGIN_NONE]
    seq = synth.SyntheticSequence(23, n_frames=13, K=12, L=24)
    yield "K=12", [seq.window(0)], [ba.VG_MARGIN_NONE]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ba_gpu import relocalisation_problem
    yield "relocalisation", [relocalisation_problem()], [ba.VG_MARGIN_OLD]


def run(h, probs, flags, large=False):
    h.ba_set_large_window(large)
    try:
        h.ba_upload(probs, flags)
        h.ba_run_async()
        return h.ba_download()
    finally:
        h.ba_set_large_window(False)


def same(a, b, what):
    if isinstance(a, dict):
        for k in a:
            same(a[k], b[k], f"{what}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), what
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{what}[{i}]")
    elif a is None or b is None:
        assert a is None and b is None, what
    else:
        x, y = np.asarray(a), np.asarray(b)
        if x.dtype.kind == 'f':
            ok = np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y)
        else:
            ok = np.array_equal(x, y)
        if 'prof' in what:
            return
        assert ok, f"{what}: differs (max abs {np.nanmax(np.abs(x.astype(float) - y.astype(float)))})"


if __name__ == "__main__":
    ha, hb = handle_of(sys.argv[1]), handle_of(sys.argv[2])
    only = sys.argv[3] if len(sys.argv) > 3 else None
    for name, probs, flags in cases():
        if only and only not in name:
            continue
        for large in (False, True):
            if large and any(f != ba.VG_MARGIN_NONE for f in flags) and len(probs) > 1:
                continue
            ra, rb = run(ha, probs, flags, large), run(hb, probs, flags, large)
            same(ra, rb, name)
            print("identical:", name, "(large-window path)" if large else "")
