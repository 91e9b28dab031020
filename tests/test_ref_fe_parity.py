"""The in-tree half of the FRONT-END oracle pinned against the reference's own code, on the CPU.

oracle/_ref/libvins_ref_fe.so = the reference's feature_tracker.cpp / parameters.cpp / feature_tracker_node.cpp and
camera_model/src/camera_models/*.cc compiled unchanged.  What the oracle (oracle/fe_cpu.cpp) RESTATES of that in-tree code —
PinholeCamera::liftProjective and FeatureTracker::setMask — is held to it here; the drop-in classes are held to the whole node in
tests/test_fe_dropin.py.  (The five OpenCV algorithms behind the reference's cv:: calls remain PARITY UNPINNED: OpenCV is absent;
tests/golden/make_golden_opencv.py is the kit that pins them where it is not.)"""
import numpy as np
import pytest

from oracle import fe_cpu as F
from oracle import ref_fe as RF

pytestmark = pytest.mark.skipif(not RF.available("ref"), reason="oracle/_ref front-end library is not built")
EUROC = (4.616e+02, 4.603e+02, 3.630e+02, 2.481e+02, -2.917e-01, 8.228e-02, 5.333e-05, -1.578e-04)


@pytest.mark.parametrize("intr", [EUROC, (6.165e+02, 6.167e+02, 3.284e+02, 2.334e+02, 9.2e-02, -1.8e-01, 1.1e-03, -2.1e-03),
                                  (500.0, 500.0, 320.0, 240.0, 0.0, 0.0, 0.0, 0.0)])
def test_lift_projective_restatement_equals_the_reference_camera(tmp_path, intr):
    """oracle_fe_lift vs camodocal::PinholeCamera::liftProjective as the node loads it from a configuration file (CameraFactory ->
    PinholeCamera::Parameters::readFromYamlFile -> setParameters: m_inv_K*, m_noDistortion for the all-zero camera)."""
    cfg = RF.write_config(str(tmp_path / "c.yaml"), intr=intr[:4], dist=intr[4:])
    node = RF.Node(RF.lib(), cfg)
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.uniform([-20, -20], [772, 500], (3000, 2)), [[0, 0], [751, 479], [363.0, 248.1]]]).astype(np.float32)
    P = node.lift(pts.astype(np.float64))
    assert np.all(P[:, 2] == 1.0)
    want = np.stack([P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]], 1).astype(np.float32)          # cv::Point2f(b.x() / b.z(), ..) (feature_tracker.cpp:266)
    got = F.lift(pts, intr)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _grid_points(rng, n, w=752, h=480):
    return np.stack([rng.uniform(2, w - 3, n), rng.uniform(2, h - 3, n)], 1).astype(np.float32)


def test_set_mask_restatement_equals_the_reference_without_ties(tmp_path):
    """FeatureTracker::setMask (feature_tracker.cpp:36-69) with pairwise different track counts: the order is unambiguous, kept tracks
    and the final mask (cv::circle stand-in = the midpoint loop of drawing.cpp; oracle = {dx^2 + dy^2 <= r^2}) must agree exactly."""
    node = RF.Node(RF.lib(), RF.write_config(str(tmp_path / "c.yaml")))
    rng = np.random.default_rng(8)
    for n in (1, 40, 150, 400):
        pts = _grid_points(rng, n)
        cnt = rng.permutation(n).astype(np.int32) + 1
        ids = np.arange(n, dtype=np.int32) + 7
        po, io, co = node.set_mask(pts, ids, cnt)
        kept, mask = F.setmask(pts, cnt, 752, 480, 30)
        assert np.array_equal(io, ids[kept]) and np.array_equal(co, cnt[kept]) and np.array_equal(po, pts[kept])
        assert np.array_equal(node.mask(), mask)
        assert len(kept) < n or n == 1


def test_set_mask_ties_follow_the_platform_sort(tmp_path):
    """With equal counts the reference's order is std::sort's (introsort: insertion sort, i.e. stable, up to 16 elements; unstable
    beyond).  The oracle canonicalises to a stable order (ASSUMPTIONS F7): equal for n <= 16, and for larger n both walks are valid
    outcomes of the same rule — every kept point is 255-clear of the points kept before it, nothing else could have been kept.  The
    drop-ins therefore call std::sort themselves and hand the device that order (tests/test_fe_dropin.py holds them to the node)."""
    node = RF.Node(RF.lib(), RF.write_config(str(tmp_path / "c.yaml")))
    rng = np.random.default_rng(9)
    pts = _grid_points(rng, 16)
    cnt = rng.integers(2, 5, 16).astype(np.int32)
    ids = np.arange(16, dtype=np.int32)
    _, io, _ = node.set_mask(pts, ids, cnt)
    kept, _ = F.setmask(pts, cnt, 752, 480, 30)
    assert np.array_equal(io, ids[kept])
    pts = _grid_points(rng, 200)
    cnt = rng.integers(2, 6, 200).astype(np.int32)
    ids = np.arange(200, dtype=np.int32)
    po, io, co = node.set_mask(pts, ids, cnt)
    assert np.all(np.diff(co) <= 0)                                    # descending counts
    r = np.rint(po).astype(int)
    for k in range(1, len(r)):                                         # kept points respect MIN_DIST among themselves
        assert ((r[:k] - r[k]) ** 2).sum(1).min() > 30 ** 2
    rest = np.setdiff1d(ids, io)
    rr = np.rint(pts[rest]).astype(int)
    assert all(((r - q) ** 2).sum(1).min() <= 30 ** 2 for q in rr)     # every dropped point lies inside a kept disc


def test_frequency_gate_and_first_frames_of_the_reference_node(tmp_path):
    """What the stand-alone harness has to reproduce when it sets PUB_THIS_FRAME itself (feature_tracker_node.cpp:29-62, :160-165): the
    first image only arms the clock, the first published cloud is withheld, FREQ 10 on a 20 Hz stream publishes every other frame."""
    import fe_scene
    frames = fe_scene.moving_scene(9, seed=2, patch=False)
    node = RF.Node(RF.lib(), RF.write_config(str(tmp_path / "c.yaml")))
    tr = [node.image(5.0 + 0.05 * k, f) for k, f in enumerate(frames)]
    assert [t['pub'] for t in tr] == [False, False, True, False, True, False, True, False, True]
    assert len(tr[0]['ids']) == 0 and len(tr[1]['ids']) == 0 and len(tr[2]['ids']) == 150
    pub = node.published()
    assert len(pub) == 3 and pub[0][0] == pytest.approx(5.2)                    # frames 4, 6, 8: the cloud of frame 2 is dropped (init_pub)
    ids = pub[0][1][:, 3].astype(int)
    assert np.array_equal(ids, tr[4]['ids'][tr[4]['track_cnt'] > 1])            # only tracks seen at least twice go out
    assert np.all(pub[0][1][:, 2] == 1.0)
