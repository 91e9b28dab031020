"""The kernel sources under the CPU fiber emulator AND AddressSanitizer (race detection / sanitizers).  In the
emulated build every device buffer is a heap block, so an out-of-bounds read or write of a kernel -- or of the host code that packs,
uploads and unpacks around it, including writes into caller-owned (NumPy) buffers -- is a reported heap-buffer-overflow instead of a
silent corruption.  (The one host overflow of round 3, a state download that wrote more inverse depths than the caller's buffer
held, was found on the GPU as a crash three tests later; this run reports it at the memcpy.)  A representative subset runs here on
every `not gpu` pass; `tests/simt/README.md` has the command for all of them (37 tests, 2.5 min)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT = os.path.join(ROOT, "tests", "simt")
# the sanitizer build lives OUTSIDE the repository (the GPU pool refuses repository snapshots that carry -fsanitize=address objects)
import tempfile  # noqa: E402
ASAN_OUT = os.path.join(tempfile.gettempdir(), "vins_simt_build_asan_%d" % os.getuid())


def _asan_runtime():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def test_emulated_kernels_under_address_sanitizer():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libasan in this toolchain")
    b = subprocess.run(["make", "-C", SIMT, "-j", str(os.cpu_count() or 4), "asan", "ASAN_OUT=" + ASAN_OUT], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0",
               VINS_SIMT_LIB=os.path.join(ASAN_OUT, "libvinsgpu_simt.so"))
    sel = ("test_emulated_solve_matches_oracle or test_emulated_marginalization_matches_oracle or test_emulated_large_window_path_matches_oracle "
           "or test_emulated_enlarged_window_marginalization or test_resident_sequence_equals_host_bookkeeping or test_hand_back_and_reseed "
           "or test_emulated_detection_path_is_bit_exact_in_every_fiber_order or test_emulated_lk_is_bit_exact_in_every_fiber_order")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_simt_ba.py"), os.path.join(ROOT, "tests", "test_seq_simt.py"),
                        os.path.join(ROOT, "tests", "test_simt_fe.py"), "-q", "-x", "-p", "no:cacheprovider", "-k", sel, "--deselect",
                        "tests/test_simt_fe.py::test_emulated_lk_is_bit_exact_in_every_fiber_order[reverse]", "--deselect",
                        "tests/test_simt_fe.py::test_emulated_lk_is_bit_exact_in_every_fiber_order[shuffle]"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stdout + r.stderr, tail
    assert r.returncode == 0 and " passed" in r.stdout, tail


def test_cpp_host_side_under_address_and_leak_sanitizer(tmp_path, monkeypatch):
    """The C++ host side (Estimator / FeatureTracker shims, ResidentEstimators incl. handBack / reseed, the replay harness) built with
    -fsanitize=address against the sanitized emulated library: `vins_replay seq` (plain and with a hand-back), `ba`, `fe` and `vio`, with the
    leak checker on."""
    import struct
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import replay_util
    import seq_model as M
    from vins_mono_amd import synth
    if _asan_runtime() is None:
        pytest.skip("no libasan in this toolchain")
    b = subprocess.run(["make", "-C", SIMT, "-j", str(os.cpu_count() or 4), "asan", "ASAN_OUT=" + ASAN_OUT], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    exe = os.path.join(ASAN_OUT, "vins_replay_simt")
    K, n_frames = 11, 3
    src = [M.FrameSource(synth.SyntheticSequence(s, n_frames=K + n_frames + 1, K=K + n_frames + 1, L=70), noise_seed=200 + s) for s in (21, 22)]
    M.write_seq_file(tmp_path / "frames.bin", src, K, n_frames, min_parallax=0.25)
    replay_util.write_sequence(replay_util.make_plan(5, 3, L=40), str(tmp_path / "seq.bin"))
    W, H = 376, 240
    frames = [synth.synth_frame(3, W, H)]
    for k in range(1, 3):
        frames.append(synth.warp_frame(frames[-1], 10 + k, shift=(2.1, -1.3), angle_deg=0.4))
    with open(tmp_path / "fe.bin", "wb") as f:
        f.write(struct.pack("<4i", len(frames), W, H, 1))
        for fr in frames:
            f.write(np.ascontiguousarray(fr).tobytes())
    runs = [(["seq", "frames.bin"], {}), (["seq", "frames.bin"], {"VINS_REPLAY_HANDBACK": "0"}), (["ba", "seq.bin"], {}), (["fe", "fe.bin"], {})]
    for args, extra in runs:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:detect_stack_use_after_return=0", **extra)
        r = subprocess.run([exe, args[0], str(tmp_path / args[1]), str(tmp_path / "out.txt")], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, (args, extra, r.stderr[-3000:])
    # `vio`: both drop-ins in one process on rendered frames (tests/e2e_vio.py), the window's ten frames + two solved ones
    import e2e_vio
    scene = e2e_vio.Scene(3, 12)
    monkeypatch.setenv("ASAN_OPTIONS", "detect_leaks=1:detect_stack_use_after_return=0")
    out = e2e_vio.run_vio_replay(exe, scene, [scene.render(f) for f in range(12)], str(tmp_path))
    assert len(out) == 2 and all(o[4] == 0 for o in out), out
