"""The front-end drop-in inside the REFERENCE'S OWN NODE.

oracle/_ref/libvins_ref_fe.so is the reference's feature_tracker_node.cpp, feature_tracker.cpp, parameters.cpp and camera_model/src/
camera_models/*.cc compiled unchanged (oracle/Makefile: ref_fe) on header stand-ins whose five cv:: algorithm calls forward to
oracle/fe_cpu.cpp.  libvins_ref_fe_gpu.so / libvins_ref_fe_simt.so are the SAME objects with four members of `class FeatureTracker`
— readImage, setMask, rejectWithF, undistortedPoints — replaced by vins-mono_amd/host/dropin/feature_tracker_readimage.cpp, which is
written against the reference's own feature_tracker.h (camodocal::CameraPtr m_camera and all) and calls the C-ABI of libvinsgpu.so
(_simt: the same kernel sources on the CPU emulator, so this also runs in the `not gpu` suite).

Both builds receive the same sensor_msgs/Image stream through the node's img_callback(): first-image handling, the PUB_THIS_FRAME
frequency gate (feature_tracker_node.cpp:29-62), readImage, updateID (:103-111), the `feature` point cloud (:113-163).  Nothing of
the reference is restated: what differs between the two runs is exactly the code this repo replaces.

Acceptance = BASELINE.json north_star "bit-exact feature indices": per frame identical ids, track_cnt, cur_pts, cur_un_pts,
pts_velocity (float bit patterns) and identical published clouds."""
import numpy as np
import pytest

import fe_scene
from oracle import ref_fe as RF

needs_ref = pytest.mark.skipif(not RF.available("ref"), reason="oracle/_ref front-end libraries are not built")


def _default_config():
    """config/euroc/euroc_config.yaml's front-end keys (RF.write_config defaults), written where the GPU box can read it too"""
    import os
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "vins_fe_euroc_%d.yaml" % os.getpid())
    return RF.write_config(path)


LAST_STATS = {}


def _run(L, frames, config=None, stamps=None, options=(), vins_folder=""):
    node = RF.Node(L, config or _default_config(), vins_folder=vins_folder)
    for k, v in options:
        node.set_option(k, v)
    out = []
    for k, f in enumerate(frames):
        out.append(node.image(100.0 + 0.05 * k if stamps is None else stamps[k], f))
    LAST_STATS.clear(); LAST_STATS.update(node.gpu_stats())
    return out, node.published(), node.restarts()


def _one_call_per_frame(n_frames):
    """the drop-in run that just ended went through vg_fe_read_image on every frame, rejectWithF ran on the device (no estimate was
    handed back to the host) and its bookkeeping stopped the iterations early, as OpenCV's loop does"""
    s = LAST_STATS
    assert s["frames"] == n_frames - 1 and s["stepwise"] == 0 and 0 < s["published"] < n_frames, s      # (the node drops the first image, :33-39)
    assert s["ransac"] >= s["published"] - 1 and s["fb_collinear"] == 0 and s["fb_lmeds"] == 0, s
    assert 0 < s["niters"] < 1000 * s["ransac"], s
    print("drop-in frames:", s)


def _write_fisheye_mask(folder, mask):
    """<vins_folder>config/fisheye_mask.jpg (feature_tracker/src/parameters.cpp:62) as a binary PGM: what the cv::imread stand-in reads"""
    import os
    os.makedirs(os.path.join(folder, "config"), exist_ok=True)
    with open(os.path.join(folder, "config", "fisheye_mask.jpg"), "wb") as f:
        f.write(b"P5 %d %d 255\n" % (mask.shape[1], mask.shape[0]) + mask.tobytes())
    return folder + "/"


def _compare(ref, got):
    (ra, rp, rr), (ga, gp, gr) = ref, got
    assert len(ra) == len(ga) and rr == gr
    for k, (a, b) in enumerate(zip(ra, ga)):
        assert a['pub'] == b['pub'] and a['n_id'] == b['n_id'], k
        assert np.array_equal(a['ids'], b['ids']), k
        assert np.array_equal(a['track_cnt'], b['track_cnt']), k
        for key in ('cur_pts', 'cur_un_pts', 'pts_velocity'):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (k, key)
    assert len(rp) == len(gp)
    for (sa, ca), (sb, cb) in zip(rp, gp):
        assert sa == sb and np.array_equal(ca.view(np.uint32), cb.view(np.uint32))


def _interesting(run, n_frames):
    """the stream exercised what it should: published and unpublished frames, lost tracks, new ids after the first detection, long tracks"""
    tr, pub, _ = run
    pubs = [t['pub'] for t in tr]
    assert any(pubs) and not all(pubs[2:])
    assert tr[-1]['n_id'] > 200 and len(tr[-1]['ids']) >= 100
    assert max(t['track_cnt'].max() for t in tr if len(t['ids'])) >= 8
    assert len(pub) >= (n_frames - 4) // 2 - 1
    assert any(len(a['ids']) < len(b['ids']) for a, b in zip(tr[3:], tr[2:]))          # LK / border / RANSAC drop tracks


@needs_ref
@pytest.mark.skipif(not RF.available("simt"), reason="emulated drop-in library is not built")
def test_reference_node_with_the_drop_in_on_emulated_kernels():
    """24 frames at 752x480, EuRoC configuration (CLAHE on, MAX_CNT 150, MIN_DIST 30, FREQ 10 at a 20 Hz stream)."""
    frames = fe_scene.moving_scene(24)
    ref = _run(RF.lib(), frames)
    _interesting(ref, 24)
    _compare(ref, _run(RF.lib_simt(), frames))
    _one_call_per_frame(24)


@needs_ref
@pytest.mark.skipif(not RF.available("simt"), reason="emulated drop-in library is not built")
def test_stream_discontinuity_and_fisheye_mask_on_emulated_kernels(tmp_path):
    """A time jump > 1 s makes the node publish `restart` and re-arm its first-image logic while the tracker keeps its tracks
    (feature_tracker_node.cpp:37-48); fisheye: 1 starts setMask() from the fisheye mask instead of all-255 (feature_tracker.cpp:38-41)."""
    frames = fe_scene.moving_scene(12, seed=9, velocity=(-3.1, 2.6))
    stamps = [50.0 + 0.05 * k for k in range(6)] + [60.0 + 0.05 * k for k in range(6)]
    cfg = RF.write_config(str(tmp_path / "cfg.yaml"), equalize=0, fisheye=1, max_cnt=120, min_dist=25)
    yy, xx = np.mgrid[0:480, 0:752]
    mask = np.where((xx - 376) ** 2 + (yy - 240) ** 2 < 300 ** 2, 255, 0).astype(np.uint8)
    folder = _write_fisheye_mask(str(tmp_path), mask)
    ref = _run(RF.lib(), frames, cfg, stamps, vins_folder=folder)
    assert ref[2] == 1                                                          # one restart message
    pts = np.concatenate([t['cur_pts'] for t in ref[0] if t['pub'] and len(t['ids'])])
    assert len(pts) and ((pts[:, 0] - 376) ** 2 + (pts[:, 1] - 240) ** 2 < 331 ** 2).all()      # nothing detected outside the fisheye mask (+ LK drift)
    _compare(ref, _run(RF.lib_simt(), frames, cfg, stamps, vins_folder=folder))


@needs_ref
@pytest.mark.skipif(not RF.available("simt"), reason="emulated drop-in library is not built")
def test_other_camera_model_takes_the_step_by_step_members_on_emulated_kernels(tmp_path):
    """A camera that is not a camodocal::PinholeCamera (here the unified MEI model, CataCamera.cc) is lifted by camodocal on the host, so
    readImage runs its step-by-step form -- vg_fe_push_frames / vg_fe_track / rejectWithF() / setMask() / vg_fe_detect_masked /
    undistortedPoints() -- instead of vg_fe_read_image: still bit-identical to the reference's class with the same camera."""
    frames = fe_scene.moving_scene(10, seed=17, width=320, height=240, velocity=(2.4, -1.1))
    cfg = RF.write_config(str(tmp_path / "cfg.yaml"), width=320, height=240, max_cnt=60, min_dist=16, equalize=1, freq=10,
                          intr=(310.0, 309.0, 158.0, 121.5), dist=(-0.11, 0.04, 2e-4, -1e-4), mei_xi=0.9)
    ref = _run(RF.lib(), frames, cfg)
    assert any(t['pub'] for t in ref[0]) and len(ref[0][-1]['ids']) >= 30
    _compare(ref, _run(RF.lib_simt(), frames, cfg))
    assert LAST_STATS["stepwise"] == 9 and LAST_STATS["frames"] == 0, LAST_STATS


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not RF.available("gpu"), reason="drop-in library is not built")
def test_other_camera_model_takes_the_step_by_step_members_on_the_gpu(tmp_path):
    """the same on the device, at 752x480 over 24 frames"""
    frames = fe_scene.moving_scene(24, seed=18)
    cfg = RF.write_config(str(tmp_path / "cfg.yaml"), intr=(730.0, 728.0, 371.0, 243.5), dist=(-0.11, 0.04, 2e-4, -1e-4), mei_xi=0.9)
    ref = _run(RF.lib(), frames, cfg)
    _compare(ref, _run(RF.lib_gpu(), frames, cfg))
    assert LAST_STATS["stepwise"] == 23 and LAST_STATS["frames"] == 0, LAST_STATS


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not RF.available("gpu"), reason="drop-in library is not built")
@pytest.mark.parametrize("equalize,freq", [(1, 10), (0, 20)])
def test_reference_node_with_the_drop_in_on_the_gpu(tmp_path, equalize, freq):
    """40 frames at 752x480 through both builds; FREQ 20 publishes (almost) every frame of the 20 Hz stream."""
    frames = fe_scene.moving_scene(40, seed=21 + equalize)
    cfg = RF.write_config(str(tmp_path / "cfg.yaml"), equalize=equalize, freq=freq)
    ref = _run(RF.lib(), frames, cfg)
    if freq == 10:
        _interesting(ref, 40)
    _compare(ref, _run(RF.lib_gpu(), frames, cfg))
    if freq == 10:
        _one_call_per_frame(40)


@needs_ref
@pytest.mark.gpu
@pytest.mark.skipif(not RF.available("gpu"), reason="drop-in library is not built")
def test_other_image_size_camera_and_limits_on_the_gpu(tmp_path):
    """640x480 with another pinhole camera (config/realsense-like intrinsics), MAX_CNT 200, MIN_DIST 20, a fisheye mask, a time jump."""
    frames = fe_scene.moving_scene(20, seed=33, width=640, height=480, velocity=(2.2, 4.4))
    cfg = RF.write_config(str(tmp_path / "cfg.yaml"), width=640, height=480, max_cnt=200, min_dist=20, equalize=1, fisheye=1,
                          intr=(6.165e+02, 6.167e+02, 3.284e+02, 2.334e+02), dist=(9.2e-02, -1.8e-01, 1.1e-03, -2.1e-03))
    yy, xx = np.mgrid[0:480, 0:640]
    mask = np.where((xx - 320) ** 2 + (yy - 240) ** 2 < 290 ** 2, 255, 0).astype(np.uint8)
    stamps = [10.0 + 0.05 * k for k in range(9)] + [20.0 + 0.05 * k for k in range(11)]
    folder = _write_fisheye_mask(str(tmp_path), mask)
    ref = _run(RF.lib(), frames, cfg, stamps, vins_folder=folder)
    assert ref[2] == 1 and ref[0][-1]['n_id'] > 250
    _compare(ref, _run(RF.lib_gpu(), frames, cfg, stamps, vins_folder=folder))


# ---- the stand-alone class of vins-mono_amd/host (FeatureTracker look-alike used by the replay harness `vins_replay fe|vio`)
def _replay_standalone(exe, frames, tmp_path, pub_every=2):
    import struct
    import subprocess
    path, outp = tmp_path / "frames.bin", tmp_path / "out.txt"
    with open(path, "wb") as f:
        f.write(struct.pack("4i", len(frames), frames[0].shape[1], frames[0].shape[0], pub_every))
        for im in frames:
            f.write(im.tobytes())
    r = subprocess.run([exe, "fe", str(path), str(outp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = []
    for line in open(outp):
        t = line.split()
        if t[0] == "frame":
            got.append([])
        else:
            got[-1].append([float(v) for v in t])
    return [np.array(g, np.float64).reshape(-1, 8) for g in got]


def _check_standalone(exe, tmp_path, n):
    frames = fe_scene.moving_scene(n, seed=14)
    got = _replay_standalone(exe, frames, tmp_path)
    node = RF.Node(RF.lib(), _default_config())
    assert len(got) == n
    for k, f in enumerate(frames):
        t = node.read_image(0.05 * k, f, k % 2 == 0)
        g = got[k]
        assert np.array_equal(g[:, 0].astype(np.int32), t['ids']) and np.array_equal(g[:, 1].astype(np.int32), t['track_cnt']), k
        for cols, key in ((slice(2, 4), 'cur_pts'), (slice(4, 6), 'cur_un_pts'), (slice(6, 8), 'pts_velocity')):
            assert np.array_equal(g[:, cols].astype(np.float32).view(np.uint32), t[key].view(np.uint32)), (k, key)
    assert len(got[-1]) >= 100 and got[-1][:, 1].max() >= 6


@needs_ref
def test_standalone_class_replay_equals_the_reference_tracker_on_emulated_kernels(tmp_path):
    """`vins_replay_simt fe` (host/feature_tracker.cpp on the emulated kernels) against the reference's FeatureTracker::readImage +
    updateID with the same PUB_THIS_FRAME pattern — replaces the Python mirror this harness used to be held to."""
    import os
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt", "_build", "vins_replay_simt")
    if not os.path.exists(exe):
        pytest.skip("tests/simt is not built")
    _check_standalone(exe, tmp_path, 9)


@needs_ref
@pytest.mark.gpu
def test_standalone_class_replay_equals_the_reference_tracker_on_the_gpu(tmp_path):
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vins-mono_amd", "lib", "vins_replay")
    _check_standalone(exe, tmp_path, 30)
