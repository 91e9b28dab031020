"""The front-end tracking kernels under the CPU fiber emulation (tests/simt, TEST INFRASTRUCTURE): fe_pyrdown_kernel and
fe_lk_kernel compiled from the unchanged sources, run with the fibers of a wavefront scheduled forward, in reverse and
shuffled.  The emulation has no implicit wavefront lock-step, so a missing barrier between lanes that exchange data through
LDS shows up as an order-dependent (and oracle-divergent) result.  Bit-exact against oracle/fe_cpu.cpp, like on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from oracle import fe_cpu as F
from vins_mono_amd import fe, synth
h = conftest._simt_handle()
W, H = 256, 160
a = synth.synth_frame(21, W, H)
b = synth.warp_frame(a, 22, shift=(2.6, -1.7), angle_deg=0.6)
tr = fe.FrontEnd(h, W, H, 1, 64)
tr.push_frames([a])
ref_lvl = a
for lvl in range(3):
    assert np.array_equal(tr.get_level(0, lvl), ref_lvl), lvl
    ref_lvl = F.pyrdown(ref_lvl)
tr.push_frames([b])
pts = np.concatenate([F.gftt(a, 24, 0.01, 12.0), np.array([[0.0, 0.0], [255.0, 159.0], [3.2, 150.7], [250.1, 2.5], [-3.0, 80.0]], np.float32),
                      # fractions for which the fourth bilinear weight (2^14 minus the three rounded ones) is negative
                      # (fractions in (3.05e-5, 4.58e-5); coordinates in [74, 138) where float32 resolves 7.6e-6)
                      np.array([[80 + 10 * k + f, 78 + 12 * k + f] for k in range(4) for f in (3.5e-5, 4.2e-5)], np.float32)])
nxt, st, err = tr.track(0, pts)
rn, rs, re = F.lk(a, b, pts)
assert np.array_equal(st, rs), (st, rs)
assert np.array_equal(nxt.view(np.uint32), rn.view(np.uint32))
assert np.array_equal(err.view(np.uint32), re.view(np.uint32))
assert st.sum() >= 20
tr.push_frames([b])                      # b onto itself: the same fractions in the search-window weights of the first iteration
nxt, st, err = tr.track(0, pts)
rn, rs, re = F.lk(b, b, pts)
assert np.array_equal(st, rs) and np.array_equal(nxt.view(np.uint32), rn.view(np.uint32)) and np.array_equal(err.view(np.uint32), re.view(np.uint32))
print("OK", int(st.sum()))
"""


@pytest.mark.parametrize("order", ["forward", "reverse", "shuffle"])
def test_emulated_lk_is_bit_exact_in_every_fiber_order(order):
    env = dict(os.environ, SIMT_ORDER=order)
    r = subprocess.run([sys.executable, "-c", _CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_CHILD_F = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from oracle import fe_cpu as F
from vins_mono_amd import fe
from test_fe_oracle import _two_view
h = conftest._simt_handle()
tr = fe.FrontEnd(h, 64, 64, 1, 8)
for seed, n, n_out in ((3, 60, 8), (5, 40, 0), (8, 12, 2), (9, 9, 0)):       # n >= 15: RANSAC; n < 15: LMedS
    p1, p2, out = _two_view(seed, n=n, n_out=n_out)
    st_g, F_g = tr.reject_with_f(p1, p2, 1.0)
    st_o, F_o = F.reject_with_f(p1, p2, 1.0)
    assert np.array_equal(st_g, st_o), (seed, st_g, st_o)
    s = np.abs(F_o).max()
    assert np.abs(F_g - F_o).max() < 1e-6 * s, (seed, F_g, F_o)
print("OK")
"""


def test_emulated_reject_with_f_matches_restatement():
    """vg_fe_reject_with_f (host schedule + fe_ransac7_kernel + host bookkeeping) against the sequential restatement of
    OpenCV's loop: same inlier mask, same model — RANSAC for n >= 15, LMedS below."""
    r = subprocess.run([sys.executable, "-c", _CHILD_F % dict(root=ROOT)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_CHILD_D = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from oracle import fe_cpu as F
from vins_mono_amd import fe, synth
h = conftest._simt_handle()
W, H = 256, 160
a = synth.synth_frame(31, W, H)
b = synth.warp_frame(a, 32, shift=(1.5, 2.0), angle_deg=0.3)
tr = fe.FrontEnd(h, W, H, 2, 200)
# CLAHE (fe_clahe_lut_kernel / fe_clahe_apply_kernel) + the pyramid built from the equalized frame
tr.push_frames([a, b], equalize=True)
ea, eb = F.clahe(a), F.clahe(b)
assert np.array_equal(tr.get_level(0, 0), ea) and np.array_equal(tr.get_level(1, 0), eb)
assert np.array_equal(tr.get_level(1, 1), F.pyrdown(eb))
# GFTT (fe_mineig_kernel incl. the candidates, fe_select_kernel): ordered corner lists, small and large min distance, host mask
for cam, img in ((0, ea), (1, eb)):
    for n, md in ((40, 12.0), (150, 20.0)):
        assert np.array_equal(tr.detect(cam, n, 0.01, md), F.gftt(img, n, 0.01, md)), (cam, n, md)
mask = np.full((H, W), 255, np.uint8)
yy, xx = np.mgrid[0:H, 0:W]
mask[(xx - 100) ** 2 + (yy - 70) ** 2 < 45 ** 2] = 0
assert np.array_equal(tr.detect(0, 60, 0.01, 12.0, mask), F.gftt(ea, 60, 0.01, 12.0, mask))
# setMask (fe_setmask_kernel / fe_stamp_kernel) + GFTT with the mask left on the device, undistortedPoints' lifting (fe_lift_kernel)
rng = np.random.default_rng(5)
pts = [rng.uniform([-3, -3], [W + 3, H + 3], (120, 2)).astype(np.float32),
       np.concatenate([rng.uniform([60, 40], [120, 90], (60, 2)), rng.uniform([0, 0], [W, H], (40, 2))]).astype(np.float32)]
cnts = [rng.integers(1, 6, 120), rng.integers(1, 40, 100)]
kept = tr.set_mask(pts, cnts, 12, None)
for c, img in ((0, ea), (1, eb)):
    rk, rmask = F.setmask(pts[c], cnts[c], W, H, 12, None)
    assert np.array_equal(kept[c], rk), c
    assert np.array_equal(tr.get_mask(c), rmask), c
    assert np.array_equal(tr.detect_masked(c, 50, 0.01, 12.0), F.gftt(img, 50, 0.01, 12.0, rmask)), c
intr = [461.6, 460.3, 363.0, 248.1, -0.2917, 0.08228, 5.333e-05, -1.578e-04]
p = rng.uniform([0, 0], [752, 480], (200, 2)).astype(np.float32)
assert np.array_equal(tr.undistort(p, intr).view(np.uint32), F.lift(p, intr).view(np.uint32))
# CLAHE + pyramid again (a FrontEnd re-configures the handle: last) at sizes whose CLAHE tiles are 9 x 7 pixels (odd: the byte walk of the LUT kernel, interpolation cells that start on half
# pixels) and whose pyramid levels are 72 and 36 wide (the kernel without LDS: last thread of a row patches column sw at window byte 12 / 8)
for (w2, h2, seed) in ((72, 56, 41), (104, 88, 42)):
    c2 = synth.synth_frame(seed, w2, h2)
    t2 = fe.FrontEnd(h, w2, h2, 1, 16)
    t2.push_frames([c2], equalize=True)
    e2 = F.clahe(c2)
    assert np.array_equal(t2.get_level(0, 0), e2), (w2, h2)
    assert np.array_equal(t2.get_level(0, 1), F.pyrdown(e2)), (w2, h2)
    t2.push_frames([c2], equalize=False)
    assert np.array_equal(t2.get_level(0, 1), F.pyrdown(c2)), (w2, h2)
print("OK")
"""


@pytest.mark.parametrize("order", ["forward", "reverse", "shuffle"])
def test_emulated_detection_path_is_bit_exact_in_every_fiber_order(order):
    """CLAHE, goodFeaturesToTrack (min-eigenvalue map, candidates, selection with the cell grid), setMask + stamping, liftProjective
    under the emulator: the kernels of the detection path that the tracking test above does not reach, in all three fiber orders."""
    env = dict(os.environ, SIMT_ORDER=order)
    r = subprocess.run([sys.executable, "-c", _CHILD_D % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_emulated_upload_into_the_previous_images_slot_blocks_tracking_until_the_next_build():
    """the frame slot is level 0 of its pyramid set, so an upload lands in the slot of the PREVIOUS image.  upload -> build
    -> track is the normal order; upload -> track would read the new image as the old one and is refused (VG_ERR_BAD_ARG) until a
    build has rotated the sets — instead of silently tracking against a corrupted pyramid."""
    import conftest
    from vins_mono_amd import fe, synth
    h = conftest._simt_handle()
    W, H = 256, 160
    a = synth.synth_frame(31, W, H)
    b = synth.warp_frame(a, 32)
    c = synth.warp_frame(b, 33)
    tr = fe.FrontEnd(h, W, H, 1, 32)
    d = synth.warp_frame(c, 34)
    tr.push_frames([a])                                # (the very first frame is copied into a plane of its own, not aliased)
    tr.push_frames([b])
    pts = tr.detect(0, 16, min_dist=12.0)
    tr.push_frames([c])
    ok, st, _ = tr.track(0, pts)                       # b -> c
    tr.upload_frames([d])                              # lands in b's slot = level 0 of the previous pyramid (aliased)
    with pytest.raises(RuntimeError, match="upload -> build -> track"):
        tr.track(0, pts)
    tr.build_async(False)                              # c becomes the previous image, d the current one
    nxt, st2, _ = tr.track(0, ok[st.astype(bool)])     # c -> d works
    assert st2.sum() >= 8
    h.close()
