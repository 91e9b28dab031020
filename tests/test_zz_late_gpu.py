"""The GPU tests written after the round's last GPU minute (validated on the emulated kernels only; named to run last so that a surprise
on the hardware cannot hide the measured tests behind `-x`).

End to end through both hot paths on the GPU (tests/e2e_vio.py; the emulated twin is tests/test_e2e_simt.py, same scene, same
assertions): rendered frames -> `vins_replay fe` (FeatureTracker::readImage on libvinsgpu) -> device-resident estimator window
(vg_ba_seq_*) -> trajectory against the ground truth.  The front end's output is bit-identical on both back ends
(tests/test_simt_fe.py), the estimator's to 1e-9 per solve, so the bounds are the emulated test's.  Named to run last."""
import os

import pytest

import conftest
import e2e_vio
import seq_model as M

pytestmark = pytest.mark.gpu


def test_rendered_frames_through_front_end_and_estimator(tmp_path):
    if os.environ.get("VINS_TEST_SIMT") == "1":
        pytest.skip("the emulated variant is tests/test_e2e_simt.py")
    exe = os.path.join(conftest.ROOT, "vins-mono_amd", "lib", "vins_replay")
    h = conftest.new_handle()
    try:
        r = e2e_vio.check_end_to_end(h, exe, str(tmp_path))
    finally:
        h.close()
    print(r)


def test_a_failed_window_is_isolated_and_can_be_re_seeded():
    """A window whose solve goes non-finite (a NaN in the new frame's state guess) reports VG_ERR_NUMERIC, leaves the other windows of
    the batch untouched, and is brought back with vg_ba_seq_import: re-seeded with an exported copy of its neighbour and fed the
    neighbour's frames, it reproduces the neighbour bit for bit (emulated twin: tests/test_seq_simt.py)."""
    h = conftest.new_handle()
    try:
        M.run_failure_isolation(h)
    finally:
        h.close()


@pytest.mark.parametrize("K", [4, 12])
def test_resident_sequence_at_other_window_sizes(K):
    """The smallest and the largest window the sequence kernels take, against the host bookkeeping (tests/test_seq_simt.py)."""
    if os.environ.get("VINS_TEST_SIMT") == "1":
        pytest.skip("the emulated variant is tests/test_seq_simt.py")
    a, b = conftest.new_handle(), conftest.new_handle()
    try:
        flags = M.run_both(a, b, seeds=[21], K=K, L=70, n_steps=4, min_parallax=0.25, max_features=128, check=M.check_step)
    finally:
        a.close(); b.close()
    flat = [f for fr in flags for f in fr]
    assert M.NEW in flat and M.OLD in flat


def test_tracks_on_a_moving_object_do_not_pull_the_estimate(tmp_path):
    if os.environ.get("VINS_TEST_SIMT") == "1":
        pytest.skip("the emulated variant is tests/test_e2e_simt.py")
    exe = os.path.join(conftest.ROOT, "vins-mono_amd", "lib", "vins_replay")
    h = conftest.new_handle()
    try:
        r = e2e_vio.check_moving_object(h, exe, str(tmp_path))
    finally:
        h.close()
    print(r)
