"""ctypes wrapper of oracle/_build/libfe_oracle.so (oracle/fe_cpu.cpp) — TEST INFRASTRUCTURE and the timed CPU
baseline of bench.py's front-end leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libfe_oracle.so")
_lib = None
_u8 = C.POINTER(C.c_uint8)
_f4 = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_LIB)
        _lib.oracle_fe_gftt.restype = C.c_int
        _lib.oracle_fe_gftt.argtypes = [_u8, C.c_int, C.c_int, _u8, C.c_int, C.c_double, C.c_double, _f4]
        _lib.oracle_fe_lk.argtypes = [_u8, _u8, C.c_int, C.c_int, _f4, C.c_int, C.c_int, _f4, _u8, _f4]
        _lib.oracle_fe_clahe.argtypes = [_u8, C.c_int, C.c_int, C.c_double, _u8]
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8)


def pyrdown(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().oracle_fe_pyrdown(_p8(img), w, h, _p8(out))
    return out


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().oracle_fe_scharr(_p8(img), w, h, out.ctypes.data_as(C.c_void_p))
    return out


def lk(prev, nxt, pts, max_level=3):
    prev, nxt = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    h, w = prev.shape
    out, st, err = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
    lib().oracle_fe_lk(_p8(prev), _p8(nxt), w, h, pts.ctypes.data_as(_f4), n, max_level, out.ctypes.data_as(_f4), _p8(st), err.ctypes.data_as(_f4))
    return out, st, err


def lk_mt(prev, nxt, pts, nthreads, max_level=3):
    """lk() with the points of every level spread over host threads (how OpenCV runs it); identical results."""
    prev, nxt = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    h, w = prev.shape
    out, st, err = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
    lib().oracle_fe_lk_mt(_p8(prev), _p8(nxt), w, h, pts.ctypes.data_as(_f4), n, max_level, out.ctypes.data_as(_f4), _p8(st),
                          err.ctypes.data_as(_f4), int(nthreads))
    return out, st, err


def mineig(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().oracle_fe_mineig(_p8(img), w, h, out.ctypes.data_as(_f4))
    return out


def gftt(img, max_corners, quality=0.01, min_dist=30.0, mask=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((max(max_corners, 1), 2), np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = lib().oracle_fe_gftt(_p8(img), w, h, _p8(m) if m is not None else None, int(max_corners), float(quality), float(min_dist), out.ctypes.data_as(_f4))
    return out[:n].copy()


def clahe(img, clip=3.0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros_like(img)
    rc = lib().oracle_fe_clahe(_p8(img), w, h, float(clip), _p8(out))
    assert rc == 0
    return out


def setmask(pts, track_cnt, w, h, radius, base_mask=None):
    """FeatureTracker::setMask (feature_tracker.cpp:36-69): returns (kept indices in kept order, final mask)."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    cnt = np.ascontiguousarray(track_cnt, np.int32)
    n = len(pts)
    kept = np.zeros(max(n, 1), np.int32)
    mask = np.zeros((h, w), np.uint8)
    bm = None if base_mask is None else np.ascontiguousarray(base_mask, np.uint8)
    f = lib().oracle_fe_setmask
    f.restype = C.c_int
    nk = f(pts.ctypes.data_as(_f4), cnt.ctypes.data_as(C.POINTER(C.c_int)), n, _p8(bm) if bm is not None else None,
           int(w), int(h), int(radius), kept.ctypes.data_as(C.POINTER(C.c_int)), _p8(mask))
    return kept[:nk].copy(), mask


def lift(pts, intr):
    """PinholeCamera::liftProjective with the 8-step recursive distortion model; intr = fx fy cx cy k1 k2 p1 p2."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros_like(pts)
    k = np.ascontiguousarray(intr, np.float64)
    f = lib().oracle_fe_lift
    f.restype = None
    f(pts.ctypes.data_as(_f4), len(pts), k.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(_f4))
    return out


def reject_with_f(p1, p2, threshold=1.0):
    """FeatureTracker::rejectWithF's findFundamentalMat(FM_RANSAC, threshold, 0.99), deterministic restatement
    (ASSUMPTIONS.md F9).  p1, p2: [n, 2] float32 pixel coordinates of the virtual pinhole.  Returns (status u8, F 3x3)."""
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    st = np.zeros(n, np.uint8)
    Fm = np.zeros(9)
    f = lib().oracle_fe_reject_with_f
    f.restype = C.c_int
    f(p1.ctypes.data_as(_f4), p2.ctypes.data_as(_f4), n, C.c_double(threshold), _p8(st), Fm.ctypes.data_as(C.c_void_p))
    return st, Fm.reshape(3, 3)
