"""The diff kit must keep running: tests/golden/make_golden_opencv.py and make_golden_ceres.py are the one step that
turns the eleven "parity unpinned" rows of SURVEY 8 into measured diffs on a machine that has OpenCV 3.x / Ceres 1.14 -- and they cannot
run here, so nothing notices when an interface they use moves.  These tests run BOTH generators end to end against stand-ins:
  * a stub `cv2` module whose entry points forward to oracle/fe_cpu.cpp with OpenCV's signatures (createCLAHE().apply, pyrDown, Scharr,
    cornerMinEigenVal, goodFeaturesToTrack, calcOpticalFlowPyrLK, findFundamentalMat, circle);
  * the stand-in build of the reference (oracle/_ref/libvins_ref.so, `--force-standin`) in the place of the real-Ceres build;
and then hand the files they wrote to the very pick-up tests of tests/test_golden.py (VINS_GOLDEN_OPENCV / VINS_GOLDEN_CERES).  What is
checked is the plumbing -- every field the pick-up reads is written, with the shapes and dtypes it expects, and the comparison comes out
clean when generator and oracle are the same code -- not parity: these files are NOT goldens and are written to a scratch directory."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)
import test_golden as TG  # noqa: E402
from oracle import fe_cpu as F  # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(G, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub_cv2():
    """OpenCV's Python signatures as make_golden_opencv.py uses them, on top of the oracle"""
    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "stub (oracle/fe_cpu.cpp behind OpenCV's signatures)"
    cv2.CV_16S, cv2.FM_RANSAC = 3, 8

    class _Clahe:
        def __init__(self, clip, grid):
            assert (clip, tuple(grid)) == (3.0, (8, 8))

        def apply(self, img):
            return F.clahe(img)
    cv2.createCLAHE = lambda clip, grid: _Clahe(clip, grid)
    cv2.pyrDown = lambda img: F.pyrdown(img)
    cv2.Scharr = lambda img, depth, dx, dy: F.scharr(img)[..., 0 if dx else 1]
    cv2.cornerMinEigenVal = lambda img, block, ksize=3: F.mineig(img)

    def gftt(img, max_corners, quality, min_dist, mask=None):
        c = F.gftt(img, max_corners, quality, float(min_dist), mask)
        return None if len(c) == 0 else c.reshape(-1, 1, 2)
    cv2.goodFeaturesToTrack = gftt

    def lk(a, b, pts, nxt, winSize=(21, 21), maxLevel=3):
        assert nxt is None and tuple(winSize) == (21, 21) and maxLevel == 3
        n, st, err = F.lk(a, b, np.ascontiguousarray(pts.reshape(-1, 2)))
        return n.reshape(-1, 1, 2), st.reshape(-1, 1), err.reshape(-1, 1)
    cv2.calcOpticalFlowPyrLK = lk

    def fm(p1, p2, method, thr, conf):
        assert method == cv2.FM_RANSAC and conf == 0.99
        st, Fm = F.reject_with_f(p1, p2, thr)[:2]
        return (None if Fm is None else np.asarray(Fm, float).reshape(3, 3)), st.reshape(-1, 1)
    cv2.findFundamentalMat = fm

    def circle(canvas, centre, radius, colour, thickness):
        assert thickness == -1
        yy, xx = np.mgrid[0:canvas.shape[0], 0:canvas.shape[1]]
        canvas[(xx - centre[0]) ** 2 + (yy - centre[1]) ** 2 <= radius ** 2] = colour
    cv2.circle = circle
    return cv2


def test_opencv_golden_generator_still_feeds_the_pick_up_test(tmp_path, monkeypatch):
    mk = _load("make_golden_opencv")
    monkeypatch.setattr(mk, "SEEDS", (3,))                      # (one seed: the plumbing, not the coverage)
    monkeypatch.setitem(sys.modules, "cv2", _stub_cv2())
    path = str(tmp_path / "golden_opencv_from_the_stub.npz")
    mk.main(path)
    g = np.load(path, allow_pickle=False)
    want = TG.fe_outputs(F, [3])
    missing = sorted(set(want) - set(g.files))
    assert not missing, "make_golden_opencv.py no longer writes what tests/test_golden.py compares: %s" % missing
    for k in want:
        assert np.asarray(want[k]).shape == g[k].shape and np.asarray(want[k]).dtype == g[k].dtype, k
    assert "cv_version" in g.files and [int(s) for s in g["seeds"]] == [3]
    monkeypatch.setenv("VINS_GOLDEN_OPENCV", path)
    TG.test_restated_front_end_against_real_opencv()            # the pick-up test itself, on the stub's file: must compare clean


def test_ceres_golden_generator_still_feeds_the_pick_up_test(tmp_path, monkeypatch):
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref is neither built nor buildable here")
    mk = _load("make_golden_ceres")
    full = mk.cases

    def few():
        c = full()
        keep = sorted(k for k in c if k.startswith("branch_"))[:2] + ["prior_ex0_td0"]
        return {k: c[k] for k in keep}
    monkeypatch.setattr(mk, "cases", few)
    path = str(tmp_path / "golden_ceres_from_the_stand_in.npz")
    assert mk.main(force=True, path=path) == path
    g = np.load(path, allow_pickle=False)
    assert int(g["real_ceres"]) == 0 and [str(c) for c in g["columns"]] == list(mk.COLS)
    for name in few():
        for f in ("rows", "termination", "costs", "pose", "sb", "ex", "inv_depth", "td"):
            assert name + "/" + f in g.files, (name, f)
        assert g[name + "/rows"].ndim == 2 and g[name + "/rows"].shape[1] == len(mk.COLS)
    # the pick-up test on that file (VINS_GOLDEN_CERES set: a stand-in file is accepted for this purpose only); it iterates over the
    # generator's cases(), so it is given the same reduced set
    monkeypatch.setenv("VINS_GOLDEN_CERES", path)
    monkeypatch.setattr(TG, "_ceres_cases", lambda: mk)
    TG.test_restated_minimiser_against_real_ceres()
