"""Windows that stay on the device from frame to frame (vg_ba_seq_*, csrc/ba_seq.hip), kernel sources under the CPU fiber emulator:
the device-resident sequence must take the same decisions and hold the same window as the reference's bookkeeping restated on the
host (tests/seq_model.py) feeding the ordinary C-ABI frame by frame."""
import numpy as np
import pytest

import conftest
import seq_model as M


@pytest.fixture(scope="module")
def two_handles():
    a, b = conftest._simt_handle(), conftest._simt_handle()
    yield a, b
    a.close(); b.close()


@pytest.mark.parametrize("min_parallax,want_new,seeds,n_steps", [(10.0 / 460.0, False, [21], 2), (0.25, True, [22], 3)])
def test_resident_sequence_equals_host_bookkeeping(two_handles, min_parallax, want_new, seeds, n_steps):
    h_seq, h_ref = two_handles
    flags = M.run_both(h_seq, h_ref, seeds=seeds, K=11, L=70, n_steps=n_steps, min_parallax=min_parallax, max_features=128, check=M.check_step)
    flat = [f for fr in flags for f in fr]
    assert (M.NEW in flat) == want_new


@pytest.mark.parametrize("K", [4, 5, 12])
def test_resident_sequence_at_other_window_sizes(two_handles, K):
    """The smallest window the sequence kernels take (K = 4: WINDOW_SIZE 3, only tracks anchored at frame 0 are in the problem), an
    odd one, and the largest (K = 12); both slides at each."""
    h_seq, h_ref = two_handles
    flags = M.run_both(h_seq, h_ref, seeds=[21], K=K, L=70, n_steps=4, min_parallax=0.25, max_features=128, check=M.check_step)
    flat = [f for fr in flags for f in fr]
    assert M.NEW in flat and M.OLD in flat


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_resident_sequence_under_other_fiber_orders(two_handles, monkeypatch, order):
    """The emulator has no wavefront lock-step: a missing barrier in the sequence kernels (ordered compactions, table builds that
    read what other lanes wrote) shows up as a result that depends on the order the lanes run in (tests/simt/README.md)."""
    monkeypatch.setenv("SIMT_ORDER", order)
    h_seq, h_ref = two_handles
    flags = M.run_both(h_seq, h_ref, seeds=[22], K=11, L=70, n_steps=3, min_parallax=0.25, max_features=128, check=M.check_step)
    assert M.NEW in [f for fr in flags for f in fr]


def test_resident_sequence_against_the_reference_loop(two_handles):
    """The reference's own processIMU / processImage loop (oracle/_ref) against the device-resident sequence, emulated kernels,
    five frames with non-keyframes among them (the full-length runs are tests/test_seq_gpu.py)."""
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref is not built")
    flags, worst, flips = M.run_against_reference(two_handles[0], 0.1, n_frames=15)
    assert 1 in flags


def test_cpp_resident_estimators_replay(two_handles, tmp_path):
    """The C++ host side of the sequences (vins-mono_amd/host/resident_estimator.cpp: processIMU's propagation and sample
    buffers, processImage, the batched pre-integration of running and merged intervals, the hand-over of an Estimator's window)
    through `vins_replay seq`, linked against the emulated library, against the same frames driven through the Python binding."""
    import os
    import subprocess
    from vins_mono_amd import synth
    conftest._build_simt()
    K, n_frames, mp = 11, 3, 0.25
    mk = lambda: [M.FrameSource(synth.SyntheticSequence(s, n_frames=K + n_frames + 1, K=K + n_frames + 1, L=70), noise_seed=200 + s) for s in (21, 22)]
    wins = M.write_seq_file(tmp_path / "frames.bin", mk(), K, n_frames, min_parallax=mp)
    exe = os.path.join(conftest.SIMT_DIR, "_build", "vins_replay_simt")
    r = subprocess.run([exe, "seq", str(tmp_path / "frames.bin"), str(tmp_path / "out.csv")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    src = mk()                                                # (same seeds: the generator's noise stream starts over)
    wins = [s.initial_window(K, 0) for s in src]
    expected = M.drive_sequence(two_handles[0], src, wins, K, n_frames, min_parallax=mp)
    worst = M.compare_replay_csv(tmp_path / "out.csv", expected, 2)
    flags = [e[3] for row in expected for e in row]
    assert M.NEW in flags and M.OLD in flags
    print("C++ ResidentEstimators vs the Python-driven sequence: worst difference", worst)


def test_reserved_capacities_do_not_change_the_solve(two_handles):
    """vg_ba_reserve pins the batch layout (what a sequence does for its lifetime): wider capacities change strides and grid sizes,
    not the arithmetic -- states, trace and the new prior are bit-identical."""
    from vins_mono_amd import ba, synth
    h = two_handles[1]
    seq = synth.SyntheticSequence(9, L=40)
    first = seq.window(0)
    st, sm, pr = h.ba_optimize(first, ba.VG_MARGIN_OLD)
    prob = seq.next_window(st, pr, 1)
    a = h.ba_optimize(prob, ba.VG_MARGIN_OLD)
    h.ba_reserve(max_landmarks=128, max_factors=1024, max_obs=1200, max_prior_n=91)
    try:
        b = h.ba_optimize(prob, ba.VG_MARGIN_OLD)
    finally:
        h.ba_reserve()
    for k in ('pose', 'sb', 'ex', 'inv_depth'):
        assert np.array_equal(a[0][k], b[0][k]), k
    assert a[1]['final_cost'] == b[1]['final_cost'] and np.array_equal(a[1]['it_flags'], b[1]['it_flags'])
    assert a[2]['blocks'] == b[2]['blocks'] and np.array_equal(a[2]['J0'], b[2]['J0']) and np.array_equal(a[2]['r0'], b[2]['r0'])


def test_hand_back_and_reseed(two_handles):
    """vg_ba_seq_export / vg_ba_seq_import: windows exported between two frames and taken up by a new sequence (begin and import)
    continue exactly as if they had never left the device."""
    ref, got, order = M.run_handback(two_handles[0], two_handles[1], seeds=[21, 22], n_before=2, n_after=1)
    for k, o in enumerate(order):
        for key in ('pose', 'sb', 'ex'):
            assert np.array_equal(ref[0][o][key], got[0][k][key]), (k, key, np.abs(ref[0][o][key] - got[0][k][key]).max())
        assert ref[2][o]['flag'] == got[2][k]['flag'] and ref[2][o]['n_after'] == got[2][k]['n_after']
        assert ref[1][o]['final_cost'] == got[1][k]['final_cost']


def test_cpp_hand_back_and_take_over_again(two_handles, tmp_path):
    """ResidentEstimators::handBack / handOver / reseed in the middle of a `vins_replay seq` run (VINS_REPLAY_HANDBACK): the windows go
    back into host Estimator objects (states as rotation matrices, IntegrationBase objects, f_manager.feature,
    last_marginalization_info), a new batch takes them over, and the remaining frames come out as in the uninterrupted run."""
    import os
    import subprocess
    from vins_mono_amd import synth
    conftest._build_simt()
    K, n_frames, mp = 11, 3, 0.25
    mk = lambda: [M.FrameSource(synth.SyntheticSequence(s, n_frames=K + n_frames + 1, K=K + n_frames + 1, L=70), noise_seed=200 + s) for s in (21, 22)]
    M.write_seq_file(tmp_path / "frames.bin", mk(), K, n_frames, min_parallax=mp)
    exe = os.path.join(conftest.SIMT_DIR, "_build", "vins_replay_simt")
    outs = []
    for tag, env in (("plain", {}), ("handback", {"VINS_REPLAY_HANDBACK": "0"})):
        out = tmp_path / f"{tag}.csv"
        r = subprocess.run([exe, "seq", str(tmp_path / "frames.bin"), str(out)], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l.split(',') for l in open(out).read().strip().splitlines()])
    a, b = outs
    assert len(a) == len(b) == 2 * n_frames
    worst = 0.0
    for la, lb in zip(a, b):
        assert la[0] == lb[0] and la[12:] == lb[12:]             # same estimator, same key-frame decision, tracks, status, no failure
        worst = max(worst, max(abs(float(x) - float(y)) for x, y in zip(la[2:12], lb[2:12])))
    assert worst < 1e-6, worst
    print("hand-back in the middle of the run: worst difference to the uninterrupted run", worst)


def test_frame_inputs_are_validated(two_handles):
    """A frame whose feature ids are not strictly ascending (the reference's `image` map cannot hold one) is refused on the host."""
    from vins_mono_amd import synth
    h = two_handles[0]
    src = synth.FrameSource(synth.SyntheticSequence(21, n_frames=14, K=14, L=40), noise_seed=1)
    win = src.initial_window(11, 0)
    prob, tracks = synth.sequence_inputs(win)
    h.seq_begin([prob], [tracks], max_features=128, max_new_obs=128)
    try:
        ids, rows = src.image(10)
        pose, sb = src.guess(10)
        frame = dict(pose=pose, sb=sb, imu_new=src.seq.imu[9], imu_merged=None, ids=ids[::-1].copy(), obs=rows[::-1].copy())
        with pytest.raises(RuntimeError, match="strictly ascending"):
            h.seq_step([frame])
        frame.update(ids=ids, obs=rows)
        h.seq_step([frame])                                # the sequence is still usable
        assert h.seq_info()[0]['status'] == 0
    finally:
        h.seq_end()


def test_thin_and_empty_frames(two_handles):
    """Edge cases of addFeatureCheckParallax (feature_manager.cpp:45-107): a frame that continues fewer than 20 tracks is a key frame
    whatever the parallax (:73-74), a frame without any observation, a frame of new ids only, then normal frames again -- the track
    tables, flags and solves of the device-resident window against the host bookkeeping, with the parallax threshold set so that
    ordinary frames are NOT key frames."""
    h_seq, h_ref = two_handles

    def mutate(step, w, ids, rows):
        if step == 0:
            return ids[:12], rows[:12]                     # 12 continued tracks: last_track_num < 20
        if step == 1:
            return ids[:0], rows[:0]                       # nothing seen
        if step == 2:
            return ids + 100000, rows                      # only ids nobody has seen before
        return ids, rows

    flags = M.run_both(h_seq, h_ref, seeds=[22], K=11, L=70, n_steps=5, min_parallax=0.5, max_features=256, check=M.check_step, mutate=mutate)
    flat = [f for fr in flags for f in fr]
    assert flat[:3] == [M.OLD, M.OLD, M.OLD]               # < 20 continued tracks every time


def test_failed_tracks_are_removed_at_the_slide(two_handles):
    """FeatureManager::removeFailures (feature_manager.cpp:161-171) erases every track whose solve_flag is 2 -- set by setDepth for the
    tracks of the problem (negative depth), or carried by a track that is not in the problem: windows are handed over with such
    tracks, and the first slide must drop exactly those (device = host bookkeeping, list order kept)."""
    h_seq, h_ref = two_handles
    marked = []

    def prepare(win):
        out_of_problem = [ft for ft in win.features if not win.in_problem(ft)]
        in_problem = [ft for ft in win.features if win.in_problem(ft)]
        assert len(out_of_problem) >= 3 and len(in_problem) >= 3
        for ft in out_of_problem[::2] + in_problem[:2]:
            ft['flag'] = 2
        marked.append(([ft['id'] for ft in out_of_problem[::2]], [ft['id'] for ft in in_problem[:2]]))

    survivors = {}

    def check(step, w, host, flag, dev):
        M.check_step(step, w, host, flag, dev)
        survivors.setdefault(step, set(dev['tracks']['id'].tolist()))

    M.run_both(h_seq, h_ref, seeds=[21], K=11, L=70, n_steps=2, min_parallax=10.0 / 460.0, max_features=256, check=check, prepare=prepare)
    gone, resolved = marked[0]
    still_observed = [i for i in gone if i in survivors[0]]
    assert not still_observed, still_observed               # tracks outside the problem keep their flag: removed at the first slide
    assert any(i in survivors[0] for i in resolved)         # tracks of the problem get a fresh flag from setDepth (their depth is positive)


def test_a_failed_window_is_isolated_and_can_be_re_seeded(two_handles):
    """A window whose solve goes non-finite (a NaN in the new frame's state guess) reports VG_ERR_NUMERIC, leaves the other windows of
    the batch untouched, and is brought back with vg_ba_seq_import: re-seeded with an exported copy of its neighbour and fed the
    neighbour's frames, it reproduces the neighbour bit for bit."""
    M.run_failure_isolation(two_handles[0])


def test_resident_sequence_on_the_fused_factor_kernel(two_handles):
    """the device-built tables (ba_seq_build_kernel) through ba_linacc_proj_kernel (vg_ba_set_fused_min_windows(1)) against the host
    bookkeeping on the spread kernels: same decisions and tables, states to rounding"""
    h_seq, h_ref = two_handles
    h_seq.ba_set_fused_min_windows(1)
    try:
        flags = M.run_both(h_seq, h_ref, seeds=[23], K=11, L=70, n_steps=3, min_parallax=10.0 / 460.0, max_features=128,
                           check=lambda *a: M.check_step(*a, tol=1e-7, tol_depth=1e-6))
    finally:
        h_seq.ba_set_fused_min_windows(32)
    assert M.OLD in [f for fr in flags for f in fr]
