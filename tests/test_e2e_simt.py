"""End to end through both hot paths on rendered frames, kernels under the CPU emulator (tests/e2e_vio.py): scene -> `vins_replay fe`
(FeatureTracker::readImage on the emulated library) -> device-resident estimator window (vg_ba_seq_*) -> trajectory against the
ground truth.  The GPU twin is tests/test_zz_late_gpu.py."""
import os

import conftest
import e2e_vio


def test_rendered_frames_through_front_end_and_estimator(tmp_path):
    conftest._build_simt()                                                    # `make all`: the emulated library and vins_replay_simt
    exe = os.path.join(conftest.SIMT_DIR, "_build", "vins_replay_simt")
    h = conftest._simt_handle()
    try:
        r = e2e_vio.check_end_to_end(h, exe, str(tmp_path), n_frames=16)          # (the GPU twin runs 20 frames)
    finally:
        h.close()
    print(r)


def test_tracks_on_a_moving_object_do_not_pull_the_estimate(tmp_path):
    conftest._build_simt()
    exe = os.path.join(conftest.SIMT_DIR, "_build", "vins_replay_simt")
    h = conftest._simt_handle()
    try:
        r = e2e_vio.check_moving_object(h, exe, str(tmp_path), n_frames=16)
    finally:
        h.close()
    print(r)
