#!/usr/bin/env python3
"""DIFF KIT — goldens of the REAL OpenCV for rows F1-F8 / 8(f)-3 of SURVEY.md 8(a): the five calls
FeatureTracker::readImage makes (feature_tracker/src/feature_tracker.cpp:89-91 createCLAHE(3.0, Size(8,8))->apply, :113
calcOpticalFlowPyrLK(.., Size(21,21), 3), :149 goodFeaturesToTrack(.., 0.01, MIN_DIST, mask), :191 findFundamentalMat(FM_RANSAC, F_THRESHOLD,
0.99), :66 circle(mask, pt, MIN_DIST, 0, -1)).

Cannot run in the graft image (no cv2).  Anywhere `import cv2` works (ideally OpenCV 3.3.x, what ROS Kinetic ships):

    python tests/golden/make_golden_opencv.py          # writes tests/golden/golden_opencv.npz (+ prints cv2.__version__ / build flags)

tests/test_golden.py::test_restated_front_end_against_real_opencv picks the file up when present and holds oracle/fe_cpu.cpp to it
field by field (and, with -m gpu, the HIP kernels); until then the front-end arithmetic stays PARITY UNPINNED.  Inputs are the
seeded synthetic frames of vins_mono_amd/synth.py (numpy's PCG64 stream is stable across versions; their checksums are stored and
checked on pick-up so that a differing generator cannot masquerade as an OpenCV difference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from vins_mono_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEEDS = (3, 50, 77)
LK_PARAMS = dict(winSize=(21, 21), maxLevel=3)          # criteria / flags / minEigThreshold = OpenCV's defaults, as the reference leaves them


def crc(a):
    a = np.ascontiguousarray(a).astype(np.uint64)
    return np.array([int(a.sum()), int((a * a).sum()), int((a.ravel() * (np.arange(a.size, dtype=np.uint64) % 251)).sum())], np.uint64)


def inputs(seed):
    """frame pair, detection mask (discs blanked around a few points) and a point set with outliers for the RANSAC"""
    a = synth.synth_frame(seed)
    b = synth.warp_frame(a, seed + 1)
    rng = np.random.default_rng(1000 + seed)
    mask = np.full(a.shape, 255, np.uint8)
    centres = np.stack([rng.integers(40, 712, 12), rng.integers(40, 440, 12)], 1)
    yy, xx = np.mgrid[0:a.shape[0], 0:a.shape[1]]
    for cx, cy in centres:
        mask[(xx - cx) ** 2 + (yy - cy) ** 2 <= 30 ** 2] = 0
    # 120 correspondences of a camera translating + rotating in front of random depths, pixel noise, 25 gross outliers
    n = 120
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(3, 12, n)], 1)
    th = 0.04
    Rm = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([0.3, 0.05, 0.1])
    X2 = X @ Rm.T + t
    p1 = 460.0 * X[:, :2] / X[:, 2:3] + [376.0, 240.0] + rng.normal(0, 0.3, (n, 2))
    p2 = 460.0 * X2[:, :2] / X2[:, 2:3] + [376.0, 240.0] + rng.normal(0, 0.3, (n, 2))
    p2[:25] += rng.uniform(-40, 40, (25, 2))
    return a, b, mask, centres, p1.astype(np.float32), p2.astype(np.float32)


def main(path=None):
    """path: where to write (default tests/golden/golden_opencv.npz; tests/test_diff_kit.py passes a scratch file)"""
    import cv2
    out = dict(cv_version=np.array(cv2.__version__), seeds=np.array(SEEDS))
    print("OpenCV", cv2.__version__)
    for seed in SEEDS:
        a, b, mask, centres, p1, p2 = inputs(seed)
        k = "s%d/" % seed
        out[k + "crc_a"], out[k + "crc_b"], out[k + "crc_mask"] = crc(a), crc(b), crc(mask)
        clahe = cv2.createCLAHE(3.0, (8, 8))
        ea, eb = clahe.apply(a), clahe.apply(b)
        out[k + "clahe_a"] = ea
        # pyramid levels as buildOpticalFlowPyramid forms them (pyrDown chain of the image itself)
        lvl = ea
        for i in (1, 2, 3):
            lvl = cv2.pyrDown(lvl)
            out[k + "pyr%d" % i] = lvl
        # Scharr derivative as calcSharrDeriv (un-normalised 3-10-3, int16); cv2.Scharr with BORDER_REFLECT_101 equals it in the interior
        out[k + "scharr_x"] = cv2.Scharr(ea, cv2.CV_16S, 1, 0)
        out[k + "scharr_y"] = cv2.Scharr(ea, cv2.CV_16S, 0, 1)
        out[k + "mineig"] = cv2.cornerMinEigenVal(ea, 3, ksize=3)
        for tag, mc, m, md in (("150", 150, None, 30), ("400", 400, None, 30), ("7", 7, None, 30), ("masked", 150, mask, 30), ("dense", 1000, None, 5)):
            c = cv2.goodFeaturesToTrack(ea, mc, 0.01, md, mask=m)
            out[k + "gftt_" + tag] = np.zeros((0, 2), np.float32) if c is None else c.reshape(-1, 2)
        corners = out[k + "gftt_150"]
        nxt, st, err = cv2.calcOpticalFlowPyrLK(ea, eb, corners.reshape(-1, 1, 2), None, **LK_PARAMS)
        out[k + "lk_next"], out[k + "lk_status"], out[k + "lk_err"] = nxt.reshape(-1, 2), st.ravel(), err.ravel()
        # points near / across the border and on flat texture: status 0 paths
        edge = np.array([[1.5, 2.5], [750.2, 478.1], [375.5, 0.4], [-3.0, 100.0], [760.0, 20.0]], np.float32)
        nxt, st, err = cv2.calcOpticalFlowPyrLK(ea, eb, edge.reshape(-1, 1, 2), None, **LK_PARAMS)
        out[k + "lk_edge_pts"], out[k + "lk_edge_next"], out[k + "lk_edge_status"], out[k + "lk_edge_err"] = edge, nxt.reshape(-1, 2), st.ravel(), err.ravel()
        Fm, inl = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, 1.0, 0.99)
        out[k + "fm_p1"], out[k + "fm_p2"] = p1, p2
        out[k + "fm_F"] = np.zeros((3, 3)) if Fm is None else Fm[:3]
        out[k + "fm_status"] = np.ones(len(p1), np.uint8) if inl is None else inl.ravel().astype(np.uint8)
        few = slice(30, 42)                                   # 12 points: the LMedS branch OpenCV takes below 15 (ASSUMPTIONS F9)
        Fm, inl = cv2.findFundamentalMat(p1[few], p2[few], cv2.FM_RANSAC, 1.0, 0.99)
        out[k + "fm12_status"] = np.ones(12, np.uint8) if inl is None else inl.ravel().astype(np.uint8)
        # setMask's drawing primitive
        canvas = np.full(a.shape, 255, np.uint8)
        for cx, cy in centres:
            cv2.circle(canvas, (int(cx), int(cy)), 30, 0, -1)
        out[k + "circles"] = canvas
        out[k + "circle_centres"] = centres
    path = path or os.path.join(HERE, "golden_opencv.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    try:
        import cv2  # noqa: F401
    except ImportError:
        sys.exit("cv2 is not importable here: run this script where OpenCV's Python bindings exist (see the docstring)")
    main()
