"""vg_fe_read_image (one call per frame) against the step-by-step entry points: tests/fe_read_image_case.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(seen):
    assert seen["ransac_device"] >= 6 and seen["fb_lmeds"] >= 1 and seen["fb_collinear"] >= 1 and seen["no_ransac"] >= 3, seen
    assert all(0 <= v < 1000 for v in seen["niters"]) and any(v > 0 for v in seen["niters"]), seen      # the bookkeeping cut the iterations as OpenCV's loop does (0: every point an inlier)


_CHILD = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import conftest
import fe_read_image_case as case
seen = case.run(conftest._simt_handle(), conftest._simt_handle())
print("SEEN", seen)
"""


def test_one_call_frame_equals_the_step_by_step_calls_on_emulated_kernels():
    r = subprocess.run([sys.executable, "-c", _CHILD % dict(root=ROOT)], capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0 and "SEEN" in r.stdout, r.stdout[-3000:] + r.stderr[-5000:]
    _check(eval(r.stdout[r.stdout.index("SEEN") + 4:].strip().splitlines()[0]))


@pytest.mark.gpu
def test_one_call_frame_equals_the_step_by_step_calls_on_the_gpu(handle):
    import conftest
    import fe_read_image_case as case
    other = conftest.new_handle()
    try:
        _check(case.run(handle, other, W=752, H=480, n_frames=9))
    finally:
        other.close()
