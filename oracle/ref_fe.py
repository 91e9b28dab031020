"""TEST INFRASTRUCTURE — ctypes access to oracle/_ref/libvins_ref_fe*.so: the reference's OWN front-end node
(feature_tracker/src/feature_tracker_node.cpp + feature_tracker.cpp + parameters.cpp + camera_model/src/camera_models/*.cc, compiled
unchanged from /root/reference by oracle/Makefile target `ref_fe`, driver oracle/ref_stubs_fe/ref_fe_driver.cpp).

    lib()       libvins_ref_fe.so        all reference; its five cv:: algorithm calls forward to oracle/fe_cpu.cpp (PARITY UNPINNED for
                                         those: OpenCV is absent) — readImage / setMask / rejectWithF / undistortedPoints / updateID, the
                                         PUB_THIS_FRAME gate and PinholeCamera::liftProjective are the reference's own code
    lib_gpu()   libvins_ref_fe_gpu.so    the same objects, the four FeatureTracker members replaced by the product's drop-in
                                         (vins-mono_amd/host/dropin/feature_tracker_readimage.cpp -> libvinsgpu.so); needs a GPU
    lib_simt()  libvins_ref_fe_simt.so   the drop-in on the emulated kernels (tests/simt) — runs in the `not gpu` suite

Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SRC = "/root/reference/feature_tracker/src/feature_tracker.cpp"
EUROC_CONFIG = "/root/reference/config/euroc/euroc_config.yaml"
_LIBS = {}
FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


def _path(kind):
    return os.path.join(_HERE, "_ref", {"ref": "libvins_ref_fe.so", "gpu": "libvins_ref_fe_gpu.so", "simt": "libvins_ref_fe_simt.so"}[kind])


def available(kind="ref"):
    return os.path.exists(_path(kind)) or os.path.exists(_REF_SRC)


def _load(kind):
    if kind not in _LIBS:
        if os.path.exists(_REF_SRC):                 # (re)build when the reference is present; a no-op when up to date
            if kind == "simt":
                subprocess.check_call(["make", "-C", os.path.join(_HERE, "..", "tests", "simt")], stdout=subprocess.DEVNULL)
            subprocess.check_call(["make", "-C", _HERE, "ref_simt" if kind == "simt" else "ref_fe"], stdout=subprocess.DEVNULL)
        if not os.path.exists(_path(kind)):
            raise RuntimeError(_path(kind) + " is missing and /root/reference is not here to build it")
        if kind == "simt":
            from oracle.ref import dlopen_own_scope
            L = dlopen_own_scope(_path(kind))      # (its vg_* references must bind to the emulated library: see there)
        else:
            L = C.CDLL(_path(kind))
        assert L.vfe_abi_version() == 1 and L.vfe_has_gpu_readimage() == (0 if kind == "ref" else 1)
        L.vfe_published_stamp.restype = C.c_double
        _LIBS[kind] = L
    return _LIBS[kind]


def lib():
    return _load("ref")


def lib_gpu():
    return _load("gpu")


def lib_simt():
    return _load("simt")


def write_config(path, width=752, height=480, max_cnt=150, min_dist=30, freq=10, equalize=1, fisheye=0, f_threshold=1.0,
                 intr=(4.616e+02, 4.603e+02, 3.630e+02, 2.481e+02), dist=(-2.917e-01, 8.228e-02, 5.333e-05, -1.578e-04), mei_xi=None):
    """A configuration file with the keys feature_tracker/src/parameters.cpp:45-60 and PinholeCamera::Parameters::readFromYamlFile
    (PinholeCamera.cc:144-183) read; defaults = config/euroc/euroc_config.yaml."""
    with open(path, "w") as f:
        f.write("%YAML:1.0\n\nimu_topic: \"/imu0\"\nimage_topic: \"/cam0/image_raw\"\noutput_path: \"/tmp/\"\n\n")
        if mei_xi is None:
            f.write("model_type: PINHOLE\ncamera_name: camera\nimage_width: %d\nimage_height: %d\n" % (width, height))
            f.write("distortion_parameters:\n   k1: %r\n   k2: %r\n   p1: %r\n   p2: %r\n" % tuple(float(v) for v in dist))
            f.write("projection_parameters:\n   fx: %r\n   fy: %r\n   cx: %r\n   cy: %r\n\n" % tuple(float(v) for v in intr))
        else:
            # CataCamera::Parameters::readFromYamlFile (CataCamera.cc:161-203): the unified (MEI) model -- a camera whose lifting the
            # drop-in leaves to camodocal on the host (its step-by-step members)
            f.write("model_type: MEI\ncamera_name: camera\nimage_width: %d\nimage_height: %d\n" % (width, height))
            f.write("mirror_parameters:\n   xi: %r\n" % float(mei_xi))
            f.write("distortion_parameters:\n   k1: %r\n   k2: %r\n   p1: %r\n   p2: %r\n" % tuple(float(v) for v in dist))
            f.write("projection_parameters:\n   gamma1: %r\n   gamma2: %r\n   u0: %r\n   v0: %r\n\n" % tuple(float(v) for v in intr))
        f.write("max_cnt: %d\nmin_dist: %d\nfreq: %d\nF_threshold: %r\nshow_track: 0\nequalize: %d\nfisheye: %d\n" %
                (max_cnt, min_dist, freq, float(f_threshold), equalize, fisheye))
    return path


class Node:
    """The front-end node from a fresh start: Node(L, config).image(stamp, img) = one message on IMAGE_TOPIC."""

    def __init__(self, L, config, vins_folder="", fisheye_mask=None):
        self.L = L
        rc = L.vfe_start(config.encode(), vins_folder.encode())
        assert rc == 0
        if fisheye_mask is not None:
            m = np.ascontiguousarray(fisheye_mask, np.uint8)
            L.vfe_set_fisheye_mask(m.ctypes.data_as(C.c_void_p), m.shape[1], m.shape[0])

    def gpu_stats(self):
        """vins_fe_gpu_stats of the drop-in: frames through vg_fe_read_image, published, with rejectWithF, estimates that went back to
        the host (collinear sample / LMedS range), RANSAC iterations that counted, frames through the step-by-step members"""
        v = (C.c_int * 8)()
        self.L.vfe_gpu_stats(v)
        return dict(zip(("frames", "published", "ransac", "fb_collinear", "fb_lmeds", "niters", "stepwise", "_"), list(v)))

    def set_option(self, name, value):
        self.L.vfe_set_option(name.encode(), C.c_double(float(value)))

    def image(self, stamp, img):
        img = np.ascontiguousarray(img, np.uint8)
        self.L.vfe_image(C.c_double(float(stamp)), img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], img.strides[0])
        return self.tracks()

    def read_image(self, stamp, img, pub):
        """FeatureTracker::readImage + updateID with PUB_THIS_FRAME set by the caller (no frequency gate)"""
        img = np.ascontiguousarray(img, np.uint8)
        self.L.vfe_read_image_direct(C.c_double(float(stamp)), img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], img.strides[0], int(bool(pub)))
        return self.tracks()

    def tracks(self):
        """what feature_tracker_node.cpp reads after readImage() + updateID(): dict(pub, ids, track_cnt, cur_pts, cur_un_pts, pts_velocity)"""
        L = self.L
        n = L.vfe_track_count()
        ids, cnt = np.zeros(n, np.int32), np.zeros(n, np.int32)
        pts, un, vel = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
        if n:
            L.vfe_tracks(ids.ctypes.data_as(IP), cnt.ctypes.data_as(IP), pts.ctypes.data_as(FP), un.ctypes.data_as(FP), vel.ctypes.data_as(FP))
        return dict(pub=bool(L.vfe_pub_this_frame()), ids=ids, track_cnt=cnt, cur_pts=pts, cur_un_pts=un, pts_velocity=vel, n_id=int(L.vfe_n_id()))

    def published(self):
        """every sensor_msgs/PointCloud on `feature` so far: list of (stamp, rows[n, 8] = x y z id u v vx vy)"""
        L = self.L
        out = []
        for k in range(L.vfe_published_count()):
            n = L.vfe_published_size(k)
            rows = np.zeros((n, 8), np.float32)
            if n:
                L.vfe_published(k, rows.ctypes.data_as(FP))
            out.append((float(L.vfe_published_stamp(k)), rows))
        return out

    def restarts(self):
        return int(self.L.vfe_restart_count())

    def lift(self, uv):
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        out = np.zeros((len(uv), 3))
        self.L.vfe_lift(uv.ctypes.data_as(C.POINTER(C.c_double)), len(uv), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def set_mask(self, pts, ids, cnt):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        ids, cnt = np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(cnt, np.int32)
        n = len(ids)
        po, io, co = np.zeros((n, 2), np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        k = self.L.vfe_set_mask(pts.ctypes.data_as(FP), ids.ctypes.data_as(IP), cnt.ctypes.data_as(IP), n, po.ctypes.data_as(FP), io.ctypes.data_as(IP),
                                co.ctypes.data_as(IP))
        return po[:k], io[:k], co[:k]

    def mask(self, w=752, h=480):
        out = np.zeros((h, w), np.uint8)
        return out if self.L.vfe_get_mask(out.ctypes.data_as(C.c_void_p), w, h) else None


def same_tracks(a, b):
    """identical per-frame state of two runs (bit patterns of the floats included)"""
    if a['pub'] != b['pub'] or a['n_id'] != b['n_id'] or len(a['ids']) != len(b['ids']):
        return False
    return all(np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])
               for k in ('ids', 'track_cnt', 'cur_pts', 'cur_un_pts', 'pts_velocity'))
