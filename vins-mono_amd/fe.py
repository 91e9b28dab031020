"""ctypes binding of the front-end entry points of include/vinsgpu.h (vg_fe_*); no arithmetic lives here."""
import ctypes as C

import numpy as np

_u8 = C.POINTER(C.c_uint8)
_f4 = C.POINTER(C.c_float)
_i4 = C.POINTER(C.c_int)


class FrameOut(C.Structure):
    """vg_fe_frame_out (include/vinsgpu.h)"""
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("ransac_ran", C.c_int), ("n_kept", C.c_int), ("n_new", C.c_int), ("n_final", C.c_int),
                ("status_lk", _u8), ("status_f", _u8), ("forw_xy", _f4), ("kept", _i4), ("new_xy", _f4), ("un_xy", _f4),
                ("ransac_best", C.c_int), ("ransac_niters", C.c_int), ("fallback", C.c_int)]


ORDER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(FrameOut), _i4)


class FrameIn(C.Structure):
    """vg_fe_frame_in (include/vinsgpu.h)"""
    _fields_ = [("struct_size", C.c_int), ("img", _u8), ("stride", C.c_int), ("equalize", C.c_int), ("publish", C.c_int), ("cur_xy", _f4),
                ("n", C.c_int), ("max_cnt", C.c_int), ("min_dist", C.c_int), ("quality", C.c_double), ("f_threshold", C.c_double),
                ("focal_length", C.c_double), ("intr", C.c_double * 8), ("base_mask", _u8), ("order", ORDER_FN), ("user", C.c_void_p)]


class FrontEnd:
    """`n_cams` camera streams on one vg_handle (ba.Handle)."""

    def __init__(self, handle, width, height, n_cams=1, max_points=150):
        self.hd, self.lib, self.h = handle, handle.lib, handle.h
        self.W, self.H, self.cams, self.max_pts = width, height, n_cams, max_points
        L = self.lib
        L.vg_fe_configure.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vg_fe_push_frames.argtypes = [C.c_void_p, C.POINTER(_u8), C.c_int, C.c_int]
        L.vg_fe_upload_frames.argtypes = [C.c_void_p, C.POINTER(_u8), C.c_int]
        L.vg_fe_build_async.argtypes = [C.c_void_p, C.c_int]
        L.vg_fe_select_frames.argtypes = [C.c_void_p, C.c_int]
        L.vg_fe_frame_slot.argtypes = [C.c_void_p]
        L.vg_fe_track.argtypes = [C.c_void_p, C.c_int, _f4, C.c_int, _f4, _u8, _f4]
        L.vg_fe_track_upload.argtypes = [C.c_void_p, _f4, _i4]
        L.vg_fe_track_async.argtypes = [C.c_void_p]
        L.vg_fe_track_download.argtypes = [C.c_void_p, _f4, _u8, _f4]
        L.vg_fe_detect.argtypes = [C.c_void_p, C.c_int, _u8, C.c_int, C.c_double, C.c_double, _f4, _i4]
        L.vg_fe_detect_upload.argtypes = [C.c_void_p, C.POINTER(_u8), _i4]
        L.vg_fe_detect_async.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.vg_fe_detect_download.argtypes = [C.c_void_p, _f4, _i4]
        L.vg_fe_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8, _i4, _i4]
        L.vg_fe_get_eig.argtypes = [C.c_void_p, C.c_int, _f4]
        L.vg_fe_keep_eig.argtypes = [C.c_void_p, C.c_int]
        L.vg_fe_set_mask.argtypes = [C.c_void_p, _f4, _i4, _i4, C.POINTER(_u8), C.c_int, _i4, _i4]
        L.vg_fe_detect_masked.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, _f4, _i4]
        L.vg_fe_get_mask.argtypes = [C.c_void_p, C.c_int, _u8]
        L.vg_fe_undistort.argtypes = [C.c_void_p, _f4, C.c_int, C.POINTER(C.c_double), _f4]
        L.vg_fe_reject_with_f.argtypes = [C.c_void_p, _f4, _f4, C.c_int, C.c_double, _u8, _i4, C.POINTER(C.c_double)]
        L.vg_fe_read_image.argtypes = [C.c_void_p, C.POINTER(FrameIn), C.POINTER(FrameOut)]
        self.hd._chk(L.vg_fe_configure(self.h, width, height, n_cams, max_points), "vg_fe_configure")

    def _imgs(self, frames):
        self._keep = [np.ascontiguousarray(f, np.uint8) for f in frames]
        assert len(self._keep) == self.cams and all(f.shape == (self.H, self.W) for f in self._keep)
        return (_u8 * self.cams)(*[f.ctypes.data_as(_u8) for f in self._keep])

    def push_frames(self, frames, equalize=False):
        self.hd._chk(self.lib.vg_fe_push_frames(self.h, self._imgs(frames), self.W, int(equalize)), "vg_fe_push_frames")

    def upload_frames(self, frames):
        self.hd._chk(self.lib.vg_fe_upload_frames(self.h, self._imgs(frames), self.W), "vg_fe_upload_frames")

    def frame_slot(self):
        """The frame slot the last upload went into (vg_fe_frame_slot)."""
        return int(self.lib.vg_fe_frame_slot(self.h))

    def select_frames(self, slot):
        self.hd._chk(self.lib.vg_fe_select_frames(self.h, int(slot)), "vg_fe_select_frames")

    def build_async(self, equalize=False):
        self.hd._chk(self.lib.vg_fe_build_async(self.h, int(equalize)), "vg_fe_build_async")

    def track(self, cam, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        out, st, err = np.zeros((n, 2), np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
        self.hd._chk(self.lib.vg_fe_track(self.h, cam, pts.ctypes.data_as(_f4), n, out.ctypes.data_as(_f4), st.ctypes.data_as(_u8),
                                          err.ctypes.data_as(_f4)), "vg_fe_track")
        return out, st, err

    def track_upload(self, pts_list):
        buf = np.zeros((self.cams, self.max_pts, 2), np.float32)
        n = np.zeros(self.cams, np.int32)
        for c, p in enumerate(pts_list):
            p = np.asarray(p, np.float32).reshape(-1, 2)
            buf[c, :p.shape[0]] = p
            n[c] = p.shape[0]
        self._n = n
        self.hd._chk(self.lib.vg_fe_track_upload(self.h, buf.ctypes.data_as(_f4), n.ctypes.data_as(_i4)), "vg_fe_track_upload")

    def track_async(self):
        self.hd._chk(self.lib.vg_fe_track_async(self.h), "vg_fe_track_async")

    def track_download(self):
        out = np.zeros((self.cams, self.max_pts, 2), np.float32)
        st = np.zeros((self.cams, self.max_pts), np.uint8)
        err = np.zeros((self.cams, self.max_pts), np.float32)
        self.hd._chk(self.lib.vg_fe_track_download(self.h, out.ctypes.data_as(_f4), st.ctypes.data_as(_u8), err.ctypes.data_as(_f4)), "vg_fe_track_download")
        return [(out[c, :self._n[c]], st[c, :self._n[c]], err[c, :self._n[c]]) for c in range(self.cams)]

    def detect(self, cam, max_corners, quality=0.01, min_dist=30.0, mask=None):
        out = np.zeros((max(max_corners, 1), 2), np.float32)
        n = C.c_int(0)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.hd._chk(self.lib.vg_fe_detect(self.h, cam, m.ctypes.data_as(_u8) if m is not None else None, int(max_corners), float(quality),
                                           float(min_dist), out.ctypes.data_as(_f4), C.byref(n)), "vg_fe_detect")
        return out[:n.value].copy()

    def set_mask(self, pts, track_cnt, radius, base_masks=None):
        """FeatureTracker::setMask for every stream: pts[c] (n_c x 2 float32), track_cnt[c] (n_c ints).  Returns the list
        of kept-index arrays (kept order); the final masks stay on the device for detect_masked()."""
        P = np.zeros((self.cams, self.max_pts, 2), np.float32)
        T = np.zeros((self.cams, self.max_pts), np.int32)
        n = np.zeros(self.cams, np.int32)
        for c in range(self.cams):
            p = np.ascontiguousarray(pts[c], np.float32).reshape(-1, 2)
            n[c] = len(p)
            P[c, :len(p)] = p
            T[c, :len(p)] = np.asarray(track_cnt[c], np.int32)
        bp = None
        if base_masks is not None:
            self._bkeep = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in base_masks]
            bp = (_u8 * self.cams)(*[None if m is None else m.ctypes.data_as(_u8) for m in self._bkeep])
        K = np.zeros((self.cams, self.max_pts), np.int32)
        nk = np.zeros(self.cams, np.int32)
        self.hd._chk(self.lib.vg_fe_set_mask(self.h, P.ctypes.data_as(_f4), T.ctypes.data_as(_i4), n.ctypes.data_as(_i4), bp, int(radius),
                                             K.ctypes.data_as(_i4), nk.ctypes.data_as(_i4)), "vg_fe_set_mask")
        return [K[c, :nk[c]].copy() for c in range(self.cams)]

    def detect_masked(self, cam, max_corners, quality=0.01, min_dist=30.0):
        out = np.zeros((max(max_corners, 1), 2), np.float32)
        n = C.c_int(0)
        self.hd._chk(self.lib.vg_fe_detect_masked(self.h, cam, int(max_corners), float(quality), float(min_dist), out.ctypes.data_as(_f4),
                                                  C.byref(n)), "vg_fe_detect_masked")
        return out[:n.value].copy()

    def get_mask(self, cam):
        out = np.zeros((self.H, self.W), np.uint8)
        self.hd._chk(self.lib.vg_fe_get_mask(self.h, cam, out.ctypes.data_as(_u8)), "vg_fe_get_mask")
        return out

    def undistort(self, pts, intr):
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(p)
        k = np.ascontiguousarray(intr, np.float64)
        self.hd._chk(self.lib.vg_fe_undistort(self.h, p.ctypes.data_as(_f4), len(p), k.ctypes.data_as(C.POINTER(C.c_double)),
                                              out.ctypes.data_as(_f4)), "vg_fe_undistort")
        return out

    def reject_with_f(self, p1, p2, threshold=1.0):
        """FeatureTracker::rejectWithF's findFundamentalMat(FM_RANSAC, threshold, 0.99) on the device (deterministic RANSAC).
        p1, p2: [n, 2] float32 virtual-pinhole pixel coordinates.  Returns (status u8 [n], F 3x3)."""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = p1.shape[0]
        st = np.zeros(n, np.uint8)
        Fm = np.zeros(9)
        ni = C.c_int()
        self.hd._chk(self.lib.vg_fe_reject_with_f(self.h, p1.ctypes.data_as(_f4), p2.ctypes.data_as(_f4), n, float(threshold),
                                                  st.ctypes.data_as(_u8), C.byref(ni), Fm.ctypes.data_as(C.POINTER(C.c_double))),
                     "vg_fe_reject_with_f")
        return st, Fm.reshape(3, 3)

    def read_image(self, img, cur_pts, publish, intr, max_cnt=150, min_dist=30, equalize=False, f_threshold=1.0, focal_length=460.0,
                   quality=0.01, base_mask=None, order=None):
        """vg_fe_read_image: FeatureTracker::readImage of one stream in one call.  `order(status_lk, status_f or None, forw_xy, n2)` returns
        the walk order of setMask as indices into the n2 survivors (None: the list as it stands).  Returns a dict of numpy copies."""
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.H, self.W)
        pts = np.ascontiguousarray(cur_pts, np.float32).reshape(-1, 2)
        n = pts.shape[0]
        fin = FrameIn()
        fin.struct_size = C.sizeof(FrameIn)
        fin.img = img.ctypes.data_as(_u8); fin.stride = self.W; fin.equalize = int(equalize); fin.publish = int(publish)
        fin.cur_xy = pts.ctypes.data_as(_f4) if n else None
        fin.n = n; fin.max_cnt = int(max_cnt); fin.min_dist = int(min_dist); fin.quality = float(quality)
        fin.f_threshold = float(f_threshold); fin.focal_length = float(focal_length)
        for i, v in enumerate(intr):
            fin.intr[i] = float(v)
        if base_mask is not None:
            self._base = np.ascontiguousarray(base_mask, np.uint8)
            assert self._base.shape == (self.H, self.W)
            fin.base_mask = self._base.ctypes.data_as(_u8)
        seen = {}

        def _cb(_user, after, out_order):
            a = after.contents
            st = np.ctypeslib.as_array(a.status_lk, (max(n, 1),))[:n].copy()
            sf = np.ctypeslib.as_array(a.status_f, (max(a.n1, 1),))[:a.n1].copy() if a.ransac_ran else None
            fw = np.ctypeslib.as_array(a.forw_xy, (max(n, 1), 2))[:n].copy()
            perm = np.asarray(order(st, sf, fw, a.n2), np.int32)
            seen["n2"] = a.n2
            if perm.shape != (a.n2,):
                return 1
            for q in range(a.n2):
                out_order[q] = int(perm[q])
            return 0

        cb = ORDER_FN(_cb) if order is not None else C.cast(None, ORDER_FN)
        fin.order = cb
        fo = FrameOut()
        self.hd._chk(self.lib.vg_fe_read_image(self.h, C.byref(fin), C.byref(fo)), "vg_fe_read_image")

        def arr(ptr, shape):
            m = int(np.prod(shape))
            return np.ctypeslib.as_array(ptr, shape).copy() if m and ptr else np.zeros(shape, np.float32 if len(shape) == 2 else np.int32)

        out = dict(n1=fo.n1, n2=fo.n2, ransac_ran=bool(fo.ransac_ran), n_kept=fo.n_kept, n_new=fo.n_new, n_final=fo.n_final,
                   ransac_best=fo.ransac_best, ransac_niters=fo.ransac_niters, fallback=fo.fallback)
        out["status_lk"] = arr(fo.status_lk, (n,)).astype(np.uint8)
        out["status_f"] = arr(fo.status_f, (fo.n1,)).astype(np.uint8) if fo.ransac_ran else None
        out["forw_xy"] = arr(fo.forw_xy, (n, 2))
        out["kept"] = arr(fo.kept, (fo.n_kept,)).astype(np.int32) if publish else None
        out["new_xy"] = arr(fo.new_xy, (fo.n_new, 2)) if publish else None
        out["un_xy"] = arr(fo.un_xy, (fo.n_final, 2))
        return out

    def detect_upload(self, max_corners, masks=None):
        mc = np.ascontiguousarray(max_corners, np.int32)
        mp = None
        if masks is not None:
            self._mkeep = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in masks]
            mp = (_u8 * self.cams)(*[None if m is None else m.ctypes.data_as(_u8) for m in self._mkeep])
        self.hd._chk(self.lib.vg_fe_detect_upload(self.h, mp, mc.ctypes.data_as(_i4)), "vg_fe_detect_upload")

    def detect_async(self, quality=0.01, min_dist=30.0):
        self.hd._chk(self.lib.vg_fe_detect_async(self.h, float(quality), float(min_dist)), "vg_fe_detect_async")

    def detect_download(self):
        out = np.zeros((self.cams, self.max_pts, 2), np.float32)
        n = np.zeros(self.cams, np.int32)
        self.hd._chk(self.lib.vg_fe_detect_download(self.h, out.ctypes.data_as(_f4), n.ctypes.data_as(_i4)), "vg_fe_detect_download")
        return [out[c, :n[c]].copy() for c in range(self.cams)]

    def get_level(self, cam, level, previous=False):
        w, hh = C.c_int(), C.c_int()
        buf = np.zeros(self.W * self.H, np.uint8)
        self.hd._chk(self.lib.vg_fe_get_level(self.h, cam, int(previous), level, buf.ctypes.data_as(_u8), C.byref(w), C.byref(hh)), "vg_fe_get_level")
        return buf[:w.value * hh.value].reshape(hh.value, w.value).copy()

    def keep_eig(self, on=True):
        """the min-eigenvalue map of every following detection is written to device memory too (it is an on-chip intermediate otherwise)"""
        self.hd._chk(self.lib.vg_fe_keep_eig(self.h, 1 if on else 0), "vg_fe_keep_eig")

    def get_eig(self, cam):
        out = np.zeros((self.H, self.W), np.float32)
        self.hd._chk(self.lib.vg_fe_get_eig(self.h, cam, out.ctypes.data_as(_f4)), "vg_fe_get_eig")
        return out
