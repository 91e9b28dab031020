"""The configuration reader (vins-mono_amd/host/yaml_config.h: the YAML subset of config/*/*.yaml incl. `!!opencv-matrix`,
replacing cv::FileStorage) — SURVEY.md section 5.

Two consumers read the same file: the shims' `readEstimatorParameters` / `readFeatureTrackerParameters` /
`FeatureTracker::readIntrinsicParameter` (host library), and — when oracle/_ref is present — the REFERENCE'S OWN
`readParameters()` (vins_estimator/src/parameters.cpp:42-137, compiled unchanged; its cv::FileStorage is a stand-in that
forwards to the same reader).  The file is a committed copy of the EuRoC configuration's VALUES with a scratch output
path (the reference creates OUTPUT_PATH and truncates the result file there)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "vins-mono_amd", "lib", "libvins_host.so")

CONFIG = """%YAML:1.0

#common parameters
imu_topic: "/imu0"
image_topic: "/cam0/image_raw"
output_path: "{out}"

#camera calibration
model_type: PINHOLE
camera_name: camera
image_width: 752
image_height: 480
distortion_parameters:
   k1: -2.917e-01
   k2: 8.228e-02
   p1: 5.333e-05
   p2: -1.578e-04
projection_parameters:
   fx: 4.616e+02
   fy: 4.603e+02
   cx: 3.630e+02
   cy: 2.481e+02

estimate_extrinsic: 1   # trailing comment
extrinsicRotation: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [0.0148655429818, -0.999880929698, 0.00414029679422,
           0.999557249008, 0.0149672133247, 0.025715529948,
           -0.0257744366974, 0.00375618835797, 0.999660727178]
extrinsicTranslation: !!opencv-matrix
   rows: 3
   cols: 1
   dt: d
   data: [-0.0216401454975,-0.064676986768, 0.00981073058949]

max_cnt: 150            # max feature number in feature tracking
min_dist: 30
freq: 0
F_threshold: 1.0
show_track: 1
equalize: 1
fisheye: 0

max_solver_time: 0.04  # max solver itration time (ms), to guarantee real time
max_num_iterations: 8
keyframe_parallax: 10.0 # keyframe selection threshold (pixel)

acc_n: 0.08
gyr_n: 0.004
acc_w: 0.00004
gyr_w: 2.0e-6
g_norm: 9.81007

estimate_td: 1
td: 0.013
rolling_shutter: 1
rolling_shutter_tr: 0.033
"""


def _write(tmp_path):
    out = tmp_path / "out"
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CONFIG.format(out=str(out) + "/"))
    return str(cfg), str(out)


def _host(cfg):
    lib = C.CDLL(HOST)
    out, intr = np.zeros(64), np.zeros(8)
    dp = C.POINTER(C.c_double)
    assert lib.vins_host_read_parameters(cfg.encode(), out.ctypes.data_as(dp), intr.ctypes.data_as(dp)) == 0
    lib.vins_host_result_path.restype = C.c_char_p
    lib.vins_host_imu_topic.restype = C.c_char_p
    return out, intr, lib.vins_host_result_path().decode(), lib.vins_host_imu_topic().decode()


def test_shims_read_the_configuration(tmp_path):
    cfg, outdir = _write(tmp_path)
    out, intr, result_path, imu_topic = _host(cfg)
    est = out[:30]
    assert np.allclose(est[:18], [0.04, 8, 10.0 / 460.0, 0.08, 0.00004, 0.004, 2.0e-6, 9.81007, 480, 752, 1, 5.0, 0.1, 0.1, 0.013, 1, 1, 0.033], rtol=0, atol=1e-15)
    Rm = est[18:27].reshape(3, 3)
    assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(Rm) - 1) < 1e-12     # re-orthonormalised through a quaternion (:105-106)
    assert np.allclose(Rm, [[0.0148655429818, -0.999880929698, 0.00414029679422], [0.999557249008, 0.0149672133247, 0.025715529948],
                            [-0.0257744366974, 0.00375618835797, 0.999660727178]], atol=1e-9)
    assert np.allclose(est[27:30], [-0.0216401454975, -0.064676986768, 0.00981073058949], atol=0)
    assert np.allclose(out[32:43], [150, 30, 480, 752, 100, 1.0, 1, 1, 0, 20, 460])                  # freq 0 -> 100 (feature_tracker parameters.cpp:67-68)
    assert np.allclose(intr, [461.6, 460.3, 363.0, 248.1, -2.917e-01, 8.228e-02, 5.333e-05, -1.578e-04], atol=0)
    assert result_path == outdir + "//vins_result_no_loop.csv" and imu_topic == "/imu0"
    assert not os.path.exists(outdir)                                                                # the shim forms names only


def test_malformed_files_are_refused(tmp_path):
    lib = C.CDLL(HOST)
    out, intr = np.zeros(64), np.zeros(8)
    dp = C.POINTER(C.c_double)
    bad = tmp_path / "bad.yaml"
    bad.write_text("%YAML:1.0\nextrinsicRotation: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [1, 0, 0, 0, 1]\n")
    assert lib.vins_host_read_parameters(str(bad).encode(), out.ctypes.data_as(dp), intr.ctypes.data_as(dp)) == -1
    assert lib.vins_host_read_parameters(str(tmp_path / "missing.yaml").encode(), out.ctypes.data_as(dp), intr.ctypes.data_as(dp)) == -1


@pytest.mark.skipif(not R.available(), reason="oracle/_ref is not built")
def test_reference_read_parameters_agrees(tmp_path):
    cfg, outdir = _write(tmp_path)
    L = R.lib()
    ref = np.zeros(32)
    assert L.vref_read_parameters(cfg.encode(), R._p(ref)) == 0
    out, _, result_path, imu_topic = _host(cfg)
    assert np.array_equal(ref[:18], out[:18])                       # scalars: the same strtod of the same text
    assert np.allclose(ref[18:27], out[18:27], atol=1e-15)          # extrinsic rotation through Quaterniond(R).normalized()
    assert np.array_equal(ref[27:30], out[27:30])
    L.vref_result_path.restype = C.c_char_p
    L.vref_imu_topic.restype = C.c_char_p
    assert L.vref_result_path().decode() == result_path and L.vref_imu_topic().decode() == imu_topic
    assert os.path.isdir(outdir) and os.path.exists(result_path)    # the reference creates the folder and truncates the result file
    R.configure()                                                   # (restore the library's defaults for other tests)
